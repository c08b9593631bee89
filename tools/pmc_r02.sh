#!/bin/bash
# Round-2 HBM-traffic counters (rocprofv3 counter mode, FETCH_SIZE and WRITE_SIZE in separate passes) of the
# BENCHMARKED configurations (WaveRNN: tools/pmc_wavernn_r02.sh): Tacotron configs[2] (B=32, 400 steps),
# HiFi-GAN f16 32x200 and Fre-GAN f16 8x3000.  Counter mode crashes on hipGraph replays, so the loops run
# eagerly (MBHIP_NO_GRAPH=1): same kernels, same arguments.  Summaries -> gpurun_out/pmc2_<what>_<counter>.json
exec < /dev/null
export TMPDIR=/tmp MBHIP_NO_GRAPH=1
mkdir -p gpurun_out
run() {  # name, command...
  local name=$1; shift
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc2_tmp
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d gpurun_out/pmc2_tmp -o p -- "$@" > gpurun_out/pmc2_${name}_$ctr.log 2>&1
    echo "$name $ctr rc=$?"
    python tools/pmc_summary.py gpurun_out/pmc2_tmp gpurun_out/pmc2_${name}_$ctr.json | head -8
    rm -rf gpurun_out/pmc2_tmp
  done
}
for what in ${@:-tacotron hifigan fregan}; do
  case $what in
    tacotron) run tacotron python tools/taco_run.py 1 ;;
    hifigan) run hifigan python tools/gan_run.py hifigan f16 32 200 2 ;;
    hifigan_f32) run hifigan_f32 python tools/gan_run.py hifigan f32 32 200 2 ;;
    fregan) run fregan python tools/gan_run.py fregan f16 8 3000 2 ;;
  esac
done
python tools/pmc_r02_json.py ${@:-tacotron hifigan fregan}   # MB_PMC_ROUND=r03 -> profiles/r03_pmc_<what>.json
