#!/bin/bash
# times the dense Gram kernel with parts of its non-MFMA work removed (results are wrong on purpose)
cd $GRAFT_REPO_ROOT && python tools/make_ablation.py
cp dnmf_amd/libdnmf_hip.so /tmp/lib_keep.so
for abl in 0 1 2; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DABL=$abl dnmf_amd/csrc/api_common.hip dnmf_amd/csrc/warp_gather.hip dnmf_amd/csrc/recon_image.hip dnmf_amd/csrc/warp_recon_grad.hip tools/abl_warp_gram_rhs.hip dnmf_amd/csrc/warp_gram_sparse.hip dnmf_amd/csrc/mu_temporal.hip dnmf_amd/csrc/render_frames.hip dnmf_amd/csrc/adam_epoch.hip dnmf_amd/csrc/spatial_update.hip dnmf_amd/csrc/image_iwarp.hip -o dnmf_amd/libdnmf_hip.so 2>&1 | grep -E "error" 
  echo "ABL=$abl"; python tools/run_k3.py --frames 2000 --reps 3
done
cp /tmp/lib_keep.so dnmf_amd/libdnmf_hip.so
