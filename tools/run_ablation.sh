#!/bin/bash
# times the dense Gram kernel with parts of its non-MFMA work removed (results are wrong on purpose).  The ablation
# builds go to build/ablation/ and are selected through DNMF_LIB: the product library is never touched.
cd "$GRAFT_REPO_ROOT" && python tools/make_ablation.py
SRC=$(python - <<'PY'
from dnmf_amd.build import SOURCES, CSRC
import os
print(" ".join(os.path.join(CSRC, s) for s in SOURCES if s != "warp_gram_rhs.hip"))
PY
)
mkdir -p build/ablation
for abl in 0 1 2; do
  hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -DABL=$abl $SRC tools/abl_warp_gram_rhs.hip -o build/ablation/k3_abl$abl.so -ldl 2>&1 | grep -E "error"
  echo "ABL=$abl"; DNMF_LIB=build/ablation/k3_abl$abl.so python tools/run_k3.py --frames 2000 --reps 3
done
