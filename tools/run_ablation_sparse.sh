#!/bin/bash
# timing-only: the table Gram kernel with its gathers served from LDS instead of global memory (results are wrong).
# The ablation build goes to build/ablation/ and is selected through DNMF_LIB: the product library is never touched.
cd "$GRAFT_REPO_ROOT" && python tools/make_ablation.py
python tools/run_k3.py --frames 4000 --reps 3 --sparse | tail -1
SRC=$(python - <<'PY'
from dnmf_amd.build import SOURCES, CSRC
import os
print(" ".join(os.path.join(CSRC, s) for s in SOURCES if s != "warp_gram_sparse.hip"))
PY
)
mkdir -p build/ablation
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 $SRC tools/abl_warp_gram_sparse.hip -o build/ablation/k3s_lds.so -ldl 2>&1 | grep -E "error"
echo "gathers from LDS:"; DNMF_LIB=build/ablation/k3s_lds.so python tools/run_k3.py --frames 4000 --reps 3 --sparse | tail -1
