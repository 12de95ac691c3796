"""``Demix.dNMF`` as the reference's demo.py imports it; the implementation lives in dnmf_amd."""
from dnmf_amd.Demix.dNMF import *  # noqa: F401,F403
from dnmf_amd.Demix.dNMF import (DeformableNMF, ExponentialFP, MultiChannelDNMF, NeuroPALVideoDataset, ResidentLoader,  # noqa: F401
                                 SimulatedVideoDataset, device)
