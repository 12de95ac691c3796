#!/bin/bash
# timing-only: the table Gram kernel with its gathers served from LDS instead of global memory (results are wrong)
cd $GRAFT_REPO_ROOT && python tools/make_ablation.py
cp dnmf_amd/libdnmf_hip.so /tmp/lib_keep.so
python tools/run_k3.py --frames 4000 --reps 3 --sparse | tail -1
SRC=$(python - <<'PY'
from dnmf_amd.build import SOURCES, CSRC
import os
print(" ".join(os.path.join(CSRC, s) for s in SOURCES if s != "warp_gram_sparse.hip"))
PY
)
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $SRC tools/abl_warp_gram_sparse.hip -o dnmf_amd/libdnmf_hip.so 2>&1 | grep -E "error"
echo "gathers from LDS:"; python tools/run_k3.py --frames 4000 --reps 3 --sparse | tail -1
cp /tmp/lib_keep.so dnmf_amd/libdnmf_hip.so
