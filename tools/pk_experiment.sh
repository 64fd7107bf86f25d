#!/bin/bash
# Co-residency experiment: does the silent corruption of the fp32 1x1 kernel beside a 16-bit-MFMA
# kernel follow ONE instruction?  Four builds of the library WITH packed-fp32 VALU ops (the shipped library is built
# without them), differing only in the victim's x * s + t (csrc/lns_kernels.hip, LNS_PKEXP):
#   pkexp0  the compiler's own code (SLP-vectorised v_pk_fma_f32 pair)          -- control: round 2 saw 1 024 .. 25 600 wrong words
#   pkexp1  four scalar v_fma_f32 in the victim, everything else still packed
#   pkexp2  the packed pair written by hand, destinations in fresh registers (no aliasing with the op_sel source)
#   pkexp3  the packed pair written by hand with the compiler's register assignment (2nd destination = the (s, t) pair)
# Build here (CPU container):  tools/pk_experiment.sh build        Run on the GPU box (once):  tools/pk_experiment.sh run
set -uo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
HIPCC=/opt/rocm/bin/hipcc
if [ "${1:-}" = build ]; then
  for v in 0 1 2 3; do
    V="$ROOT/build/variants/pkexp$v"; rm -rf "$V"; mkdir -p "$V/pkg"
    cp -r "$ROOT/lns-latent-neural-pde-solver_amd/csrc" "$V/pkg/csrc"; rm -f "$V/pkg/csrc/"*.o; cp -r "$ROOT/include" "$V/include"
    ( cd "$V/pkg/csrc" && D=""; [ $v -ge 1 ] && D="-DLNS_PKEXP=$v"
      $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $D -c lns_kernels.hip -o lns_kernels.o 2>&1 | grep -v "loop not unrolled\|^ *[0-9]* |\|^ *| *^\|generated" ;
      $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -c lns_train_kernels.hip -o lns_train_kernels.o &&
      $HIPCC -O3 -std=c++17 -fPIC -Wno-unused-function -c lns_model.cpp -o lns_model.o && $HIPCC -O3 -std=c++17 -fPIC -Wno-unused-function -c lns_engine.cpp -o lns_engine.o &&
      $HIPCC --offload-arch=gfx950 -shared -fPIC -o ../liblns_hip.so lns_kernels.o lns_train_kernels.o lns_model.o lns_engine.o ) || exit 1
    echo "pkexp$v: $(python3 "$ROOT/tools/check_isa.py" "$V/pkg/liblns_hip.so" --count 'v_pk_fma_f32' --count 'v_pk_mul_f32' | tr '\n' ' ')"
  done
  exit 0
fi
O="$ROOT/gpurun_out/pk_experiment.txt"; : > "$O"
for v in 0 1 2 3; do
  L="$ROOT/build/variants/pkexp$v/pkg/liblns_hip.so"
  echo "== pkexp$v" | tee -a "$O"
  LNS_HIP_LIB="$L" timeout -k 10 120 python3 "$ROOT/tools/pair_stress.py" exp 2>&1 | grep -v amdgpu.ids | tee -a "$O"
done
LNS_HIP_LIB= ; echo "== shipped library (no packed-fp32 ops)" | tee -a "$O"
timeout -k 10 120 python3 "$ROOT/tools/pair_stress.py" exp 2>&1 | grep -v amdgpu.ids | tee -a "$O"
