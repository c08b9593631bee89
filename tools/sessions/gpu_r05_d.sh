#!/bin/bash
# Round-5 session D: build variants of the headline kernel against the default library, interleaved on one box.
exec < /dev/null
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
  for v in default ${VARIANTS:-}; do
    if [ $v = default ]; then unset MBHIP_LIB; else export MBHIP_LIB=$PWD/build_variants/libmbhip_$v.so; fi
    timeout 200 python tools/wrn_ab_r05.py ${v}_d$rep quick > gpurun_out/r05_ab_${v}_d$rep.log 2>&1
    echo "$v #$rep: $(grep -h '^configs1_default' gpurun_out/r05_ab_${v}_d$rep.log | cut -c18-110) | $(grep -h '^configs1_exact' gpurun_out/r05_ab_${v}_d$rep.log | grep -o 'identical_to_default[^,]*')"
  done
done
unset MBHIP_LIB
if [ -n "${GROUPS3:-}" ]; then
  MBHIP_DIAG=wq_groups=3 timeout 200 python tools/wrn_ab_r05.py groups3 quick 2>&1 | grep '^configs1_default' | cut -c1-120
fi
