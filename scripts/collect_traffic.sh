#!/bin/bash
# HBM traffic of the level-1 smoother launches at 512^3 (run on the GPU box from the repo root):
# rocprofv3 --pmc in SEPARATE passes for FETCH_SIZE and WRITE_SIZE, kernel-trace only, as
# MI355X_MICROARCH.md (HBM / rocprofv3 section) prescribes; scripts/parse_traffic.py applies the
# gfx950 correction (FETCH_SIZE x 2 for 16-B/lane streams) and writes profiles/traffic_latest.json.
set -e
export TMPDIR=/tmp
out=gpurun_out/traffic
rm -rf $out; mkdir -p $out
for mode in general zero; do
  for ctr in FETCH_SIZE WRITE_SIZE; do
    arg=""; [ $mode = zero ] && arg="zero"
    rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $out/${mode}_$ctr -o t -- python3 scripts/run_sweeps.py 512 7 $arg > $out/${mode}_$ctr.log 2>&1
  done
done
python3 scripts/parse_traffic.py $out || true
