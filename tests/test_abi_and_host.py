"""CPU-side checks of the product: the C-ABI library loads and exports every symbol the header declares,
argument validation works without a GPU, and the host-side simulator reproduces the reference video."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, golden


@pytest.fixture(scope="module")
def lib():
    from dnmf_amd.build import build_library
    build_library()
    from dnmf_amd import _lib
    return _lib.load()


def header_functions():
    text = open(os.path.join(ROOT, "include", "dnmf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dnmf_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    from dnmf_amd import _lib
    names = header_functions()
    assert len(names) >= 12
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dnmf_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes prototype in dnmf_amd/_lib.py"
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_padding(lib):
    assert lib.dnmf_version() == 6
    assert [lib.dnmf_padded_k(k) for k in (0, 1, 10, 15, 16, 50, 100, 111, 112, 200)] == \
        [0, 16, 16, 16, 32, 64, 112, 112, 128, 208]


def test_argument_errors_are_reported_without_a_gpu(lib):
    """Validation happens before any HIP call, so it can be exercised on a CPU-only box."""
    rc = lib.dnmf_pack_footprints(None, 10, 3, None, 16, None)
    assert rc == -1 and b"NULL" in lib.dnmf_last_error()
    buf = ctypes.create_string_buffer(64)
    addr = ctypes.addressof(buf)
    rc = lib.dnmf_pack_footprints(addr, 10, 3, addr, 17, None)
    assert rc == -2 and b"Kp" in lib.dnmf_last_error()
    rc = lib.dnmf_mu_temporal(addr, addr, addr, 4, 300, 4, 1, None)
    assert rc == -3
    rc = lib.dnmf_warp_gram_rhs(addr, 16, 3, 0, 4, 4, 1, addr, 1, None, 1, addr, 16, None, addr, addr, addr, 8, None)
    assert rc in (-2, -4)  # alignment or workspace, never a launch
    assert lib.dnmf_warp_gram_rhs_workspace(262144, 100, 4000) == 4000 * 3 * 28 * 256 * 4
    # the neuron-list entry points
    assert lib.dnmf_pack_footprints_lists(addr, 4, 4, 1, 300, addr, addr, addr, addr, addr, None) == -3   # K > 256
    assert lib.dnmf_pack_footprints_lists(None, 4, 4, 1, 3, addr, addr, addr, addr, addr, None) == -1
    assert lib.dnmf_halo_row(512, 1) == 544 and lib.dnmf_halo_row(5, 3) == 32     # rows start on 128-byte lines
    assert lib.dnmf_halo_voxels(512, 512, 1) == 516 * 544 and lib.dnmf_halo_voxels(4, 5, 3) == 8 * 32
    assert lib.dnmf_lists_axis_masks_bytes(512, 512, 1, 100) == 2 * (512 + 512 + 1 + 6) * 2 * 8
    # slot tables (5 chunks per frame at B = 4000, two tables per chunk: one per launch; rounded up to 256 bytes), then one
    # 2-word list and one 16-byte descriptor per frame and tile
    slab = (4000 * 2 * 5 * 461 * 4 + 255) // 256 * 256
    assert lib.dnmf_warp_gram_rhs_lists_chunks(512, 512, 1, 4000) == 10     # tables a consumer sums per frame
    assert lib.dnmf_warp_gram_rhs_lists_chunks(512, 512, 1, 400) == 41      # short video: one launch, one table per chunk
    assert lib.dnmf_warp_gram_rhs_lists_workspace(461, 100, 512, 512, 1, 4000) == slab + 4000 * 1024 * (2 * 8 + 16)
    rc = lib.dnmf_warp_gram_rhs_lists(addr, addr, addr, addr, 5000, 3, 4, 4, 1, addr, 1, None, 1, addr, 16, None, addr, addr,
                                      addr, 1 << 20, None, None)
    assert rc == -3 and b"pattern slots" in lib.dnmf_last_error()
    rc = lib.dnmf_warp_gram_rhs_lists(addr, addr, addr, addr, 10, 3, 4, 4, 1, addr, 1, None, 1, addr, 16, None, addr, addr,
                                      addr, 8, None, None)
    assert rc == -4
    assert lib.dnmf_recon_image_lists(addr, addr, 3, 4, 4, 1, addr, 4, None, 2, addr, 16, None) == -2    # lds < 8 x 32
    assert lib.dnmf_mu_temporal_nbr(addr, addr, addr, 4, 3, 4, 1, addr, 12, None) == -3                   # NN not 8/16/32
    assert lib.dnmf_adam_epoch_workspace(1000) == 16000
    assert lib.dnmf_adam_epoch(addr, None, addr, addr, 4, 0, addr, None, 10, 1e-3, 0.9, 0.999, 1e-8, 0, addr, 8, None) == -4
    assert lib.dnmf_warp_recon_grad_workspace(512, 512, 1, 4000) == 512 * 8 + 4000 * 32 * 32 * 4 + 4000 * 4
    # C1: arguments are checked before RCCL is looked up
    assert lib.dnmf_comm_unique_id(None) == -1
    assert lib.dnmf_comm_init(None, addr, 2, 0) == -1
    handle = ctypes.c_void_p()
    assert lib.dnmf_comm_init(ctypes.byref(handle), addr, 2, 2) == -2 and b"rank 2 of 2" in lib.dnmf_last_error()
    assert lib.dnmf_allreduce_sum_f32(None, addr, 4, None) == -1
    assert lib.dnmf_comm_destroy(None) == 0


def test_product_has_no_cpu_fallback():
    """The classes need the HIP library and a GPU; on a CPU box construction raises instead of computing."""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dnmf_amd.Demix import dNMF
    with pytest.raises(Exception):
        dNMF.ExponentialFP(torch.tensor([8, 8, 2]), 2, 3, positions=torch.zeros(2, 3))
    import dnmf_amd
    src = "".join(open(os.path.join(os.path.dirname(dnmf_amd.__file__), f)).read()
                  for f in ("ops.py", "_lib.py", os.path.join("Demix", "dNMF.py")))
    assert "oracle" not in src.replace("CPU oracle", "")


def chan_of(NB, b, i):
    """Python mirror of csrc/warp_gram_rhs.hip:chan_of -- the channel held by lane slot i for block b."""
    ng4, r = divmod(NB, 4)
    if b < 4 * ng4:
        return 64 * (b // 4) + 4 * i + (b % 4)
    return 64 * ng4 + r * i + (b - 4 * ng4)


@pytest.mark.parametrize("NB", range(1, 9))
def test_gram_channel_permutation_is_a_bijection(NB):
    seen = sorted(chan_of(NB, b, i) for b in range(NB) for i in range(16))
    assert seen == list(range(16 * NB))
    assert chan_of(NB, NB - 1, 15) == 16 * NB - 1   # the frame column is the last pad channel


def test_simulator_reproduces_reference_video():
    from dnmf_amd.WUtils import Simulator as S
    g = golden("G8_simulator")
    par = {"sigma": [5, 5, .01], "ls": [10, 10, 10]}
    for snr, key in ((-120, "video"), (-20, "video_noisy")):
        torch.manual_seed(0)
        np.random.seed(0)
        video, positions, traces = S.generate_video(3, 6, g["sz"], 3, .2, snr, 'exp', 'gp', par)
        np.testing.assert_allclose(video.numpy(), g[key], rtol=2e-6, atol=1e-9)
        np.testing.assert_allclose(positions.numpy(), g["positions"], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(traces, g["traces"])
    with pytest.raises(NotImplementedError):
        S.generate_video(3, 6, [8, 8, 2], motion='sq')


def test_dataset_protocol_and_in_place_clamp():
    from dnmf_amd.Demix.dNMF import SimulatedVideoDataset
    torch.manual_seed(0)
    np.random.seed(0)
    ds = SimulatedVideoDataset(K=3, T=6, sz=torch.tensor([16, 16, 2]), shape_std=3, density=.2, bg_snr=-20,
                               traces='exp', motion='gp', motion_par={"sigma": [5, 5, .01], "ls": [10, 10, 10]})
    assert len(ds) == 6 and tuple(ds.video.shape) == (16, 16, 2, 6) and tuple(ds.positions.shape) == (3, 3, 6)
    assert float(ds.video.min()) < 0
    frame, idx = ds[2]
    assert idx == 2 and float(frame.min()) >= 0 and float(ds.video[..., 2].min()) >= 0   # clamped in the store
    assert float(ds.video[..., 3].min()) < 0
    loader = torch.utils.data.DataLoader(ds, batch_size=4, shuffle=False, num_workers=0)
    batch = next(iter(loader))
    assert tuple(batch[0].shape) == (4, 16, 16, 2) and batch[1].tolist() == [0, 1, 2, 3]


def test_index_batches_of_a_stock_dataloader_without_fetching_samples():
    """``DeformableNMF._resident_batches``: for a dataset that hands over its frames (``device_frames()``) only the index
    batches of a stock DataLoader are drawn -- the same batches, and the same draws from the global generator, as an
    ordinary pass over the loader; other loaders are left alone."""
    from dnmf_amd.Demix.dNMF import DeformableNMF

    class Frames(torch.utils.data.Dataset):
        fetched = 0

        def __init__(self, n, row):
            self.frames = torch.arange(n * row, dtype=torch.float32).reshape(n, row)

        def device_frames(self):
            return self.frames

        def __len__(self):
            return self.frames.shape[0]

        def __getitem__(self, i):
            Frames.fetched += 1
            return self.frames[i], i

    ds = Frames(11, 6)
    for shuffle in (False, True):
        torch.manual_seed(5)
        ordinary = [idx.tolist() for _, idx in torch.utils.data.DataLoader(ds, batch_size=4, shuffle=shuffle)]
        after_ordinary = float(torch.rand(1))
        Frames.fetched = 0
        torch.manual_seed(5)
        frames, batches = DeformableNMF._resident_batches(torch.utils.data.DataLoader(ds, batch_size=4, shuffle=shuffle), 6)
        assert batches == ordinary and Frames.fetched == 0 and frames is ds.frames
        assert float(torch.rand(1)) == after_ordinary
    assert [len(b) for b in batches] == [4, 4, 3]
    dropped = DeformableNMF._resident_batches(torch.utils.data.DataLoader(ds, batch_size=4, drop_last=True), 6)[1]
    assert [len(b) for b in dropped] == [4, 4]
    # not taken: a row length the model does not expect, a custom collate function, a dataset without device_frames
    assert DeformableNMF._resident_batches(torch.utils.data.DataLoader(ds, batch_size=4), 7) is None
    assert DeformableNMF._resident_batches(torch.utils.data.DataLoader(ds, batch_size=4, collate_fn=lambda b: b), 6) is None
    plain = torch.utils.data.TensorDataset(ds.frames, torch.arange(11))
    assert DeformableNMF._resident_batches(torch.utils.data.DataLoader(plain, batch_size=4), 6) is None
    assert DeformableNMF._resident_batches([(ds.frames[:4], torch.arange(4))], 6) is None


def test_neuropal_dataset_reads_mat_files(tmp_path):
    """The real-data loader of the reference (its data is not in the tree): round trip through scipy's .mat files."""
    from scipy.io import savemat
    from dnmf_amd.Demix.dNMF import NeuroPALVideoDataset
    rng = np.random.RandomState(0)
    data = rng.randn(12, 10, 20, 5)
    pos = 1 + rng.rand(4, 3, 5) * np.array([12, 10, 20])[None, :, None]
    savemat(tmp_path / "data.mat", {"data": data})
    savemat(tmp_path / "traces_n.mat", {"positions": pos, "neuron_names": np.array([["a", "b", "c", "d"]], dtype=object)})
    ds = NeuroPALVideoDataset(str(tmp_path))
    assert len(ds) == 5 and ds.video.shape == (6, 5, 2, 5) and tuple(ds.positions.shape) == (4, 3, 5)
    np.testing.assert_allclose(ds.positions[:, 0].numpy(), (pos[:, 0] - 1) / 2, rtol=1e-6)
    np.testing.assert_allclose(ds.positions[:, 2].numpy(), (pos[:, 2] - 1) / 10, rtol=1e-6)
    frame, idx = ds[3]
    assert idx == 3 and frame.min() >= 0 and frame.shape == (6, 5, 2)


def test_patch_grid_and_workspaces_of_the_position_initialiser(lib):
    """K8's host logic without a GPU: the patch grid equals the oracle's sliding_window_3d (reference
    MotionCorrect.py:1190-1221) -- windows of strides + overlaps every `strides`, the last one flush with the end --, windows
    that do not fit are refused, and the workspaces grow with the number of frames."""
    from oracle import motion_oracle as MO
    I3 = ctypes.c_int * 3
    for sz, strides, overlaps in (((48, 40, 2), (16, 12, 1), (8, 8, 1)), ((512, 512, 2), (96, 96, 1), (32, 32, 1)),
                                  ((40, 36, 5), (12, 12, 2), (8, 6, 1)), ((64, 64, 1), (24, 24, 1), (8, 8, 0))):
        ref = MO.sliding_window_3d(sz, overlaps, strides)
        dims = I3()
        NP = lib.dnmf_register_patches_grid(*sz, I3(*strides), I3(*overlaps), dims, None)
        assert NP == len(ref) and tuple(dims) == tuple(np.array(ref[-1][:3]) + 1)
        starts = (ctypes.c_int * (3 * NP))()
        assert lib.dnmf_register_patches_grid(*sz, I3(*strides), I3(*overlaps), dims, starts) == NP
        np.testing.assert_array_equal(np.array(starts).reshape(NP, 3), np.array([g[3:6] for g in ref]))
        w1 = lib.dnmf_register_patches_workspace(*sz, I3(*strides), I3(*overlaps), 1)
        w8 = lib.dnmf_register_patches_workspace(*sz, I3(*strides), I3(*overlaps), 8)
        assert 0 < w1 <= w8
        assert 0 < lib.dnmf_rigid_correct_workspace(*sz, 1) <= lib.dnmf_rigid_correct_workspace(*sz, 8)
    assert lib.dnmf_register_patches_grid(32, 32, 2, I3(24, 24, 1), I3(16, 16, 1), None, None) == 0      # 40 > 32: no window fits
    assert lib.dnmf_register_patches_workspace(32, 32, 2, I3(24, 24, 1), I3(16, 16, 1), 4) == 0
    assert lib.dnmf_rigid_correct_workspace(0, 4, 4, 1) == 0
