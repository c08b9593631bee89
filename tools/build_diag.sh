#!/bin/bash
# diagnostics build of libmbhip (intra-kernel marks): mockingbird_amd/libmbhip_diag.so
set -e
cd "$(dirname "$0")/../mockingbird_amd/csrc"
objs=""
for f in common conv1d gan rnn wavernn tacotron maximum_path; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -fno-gpu-rdc -DMB_TRACE_MARKS -c $f.hip -o /tmp/diag_$f.o &
  objs="$objs /tmp/diag_$f.o"
done
wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libmbhip_diag.so $objs
echo built ../libmbhip_diag.so
