#!/bin/bash
# rocprofv3 kernel stats of the headline step and of the PLT JumpReLU (dense regime) step
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
IFS="|" read -ra CF <<< "${CFGS:-head:|plt:--nonlinearity jumprelu --decoder plt|clt:--nonlinearity jumprelu}"
for cfg in "${CF[@]}"; do
  name=${cfg%%:*}; args=${cfg#*:}
  rm -rf gpurun_out/prof_$name
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$name -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline $args > gpurun_out/prof_$name.log 2>&1 || { tail -30 gpurun_out/prof_$name.log; exit 1; }
  echo "== $name"; tail -1 gpurun_out/prof_$name.log | cut -c1-200
  find gpurun_out/prof_$name -name '*kernel_stats.csv' | head -1 | xargs -r python3 -c "
import csv,sys
for i,r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i<32: print(r['Name'][:70].ljust(70), r['Calls'].rjust(5), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'])
"
done
