"""dnmf_amd -- MI355X-native deformable-NMF demixing: hand-written HIP kernels (csrc/) behind a C ABI
(include/dnmf_hip.h) and the Python mirror of the reference's ``Demix.dNMF`` surface."""
__version__ = "0.1.0"
