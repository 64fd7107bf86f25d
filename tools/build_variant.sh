#!/bin/bash
# Build a VARIANT of liblns_hip.so with extra kernel defines into build/variants/<name>/pkg/liblns_hip.so (for same-box
# A/B runs through LNS_HIP_LIB; never the shipped library).  Usage: tools/build_variant.sh <name> [-DFLAG ...]
set -euo pipefail
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name="$1"; shift
V="$ROOT/build/variants/$name"
rm -rf "$V"; mkdir -p "$V/pkg" "$V/tools"
cp -r "$ROOT/lns-latent-neural-pde-solver_amd/csrc" "$V/pkg/csrc"
rm -f "$V/pkg/csrc/"*.o
cp -r "$ROOT/include" "$V/include"
cp "$ROOT/tools/check_isa.py" "$V/tools/"
make -s -j3 -C "$V/pkg/csrc" DIAGFLAGS="$*" 2>&1 | grep -v "loop not unrolled\|^ *[0-9]* |\|^ *| *^\|warning generated\|warnings generated" || true
ls -la "$V/pkg/liblns_hip.so"
