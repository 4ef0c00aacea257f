#!/bin/bash
# Runs the stamped ablation builds of bench/micro/gemm8_ablate.hip (already built into bench/micro/bin, see the build lines
# in that file's header) on the FFN2 forward shape and prints one block per build: section times of a tile, shader cycles and
# clock of the K loop, kernel time.  Each binary is a bounded run of a few seconds; stops at the first failure.
#   bash bench/micro/run_energy_ablation.sh > gpurun_out/gemm8_energy_ablation.txt
cd "$(dirname "$0")/bin" || exit 1
SHAPE="131072 1024 4096"
run() { # run <label> <binary> [args...]
  local label=$1 bin=$2; shift 2
  [ -x "./$bin" ] || { echo "# $label: ./$bin not built"; return 0; }
  echo "# $label ($bin $SHAPE $*)"
  timeout -k 5 60 ./$bin $SHAPE "$@" 2>&1 | grep -v amdgpu.ids || exit 1
}
run "full kernel, one workgroup per tile" g8_e0 0
run "full kernel, persistent (256 workgroups)" g8_e0 256
run "no LDS-DMA in the K loop" g8_e0_a1 256
run "no LDS fragment reads" g8_e0_a2 256
run "no MFMA" g8_e0_a3 256
run "B fragments not read (-33 % LDS read bytes)" g8_e0_a6 256
run "8 extra fragment reads per phase (+33 %)" g8_e0_a7 256
run "v_mfma 32x32x16 instead of 16x16x32 (same flops)" g8_a5 256
run "L2 prefetch touches, distance 0 (off)" g8_t0 256
run "L2 prefetch touches, distance 2" g8_t2 256
run "L2 prefetch touches, distance 4" g8_t4 256
run "four-wave kernel (128x128 per wave, accumulators in AGPRs)" g4_e0 256
