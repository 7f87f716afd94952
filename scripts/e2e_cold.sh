#!/bin/bash
# Cold, like-for-like: S files through gmix_chain_many as ONE process (exec to exit) against the reference's own CLI
# (-Ofast) as min(S, 16) processes at once over the same files, and where the process's time goes.
#   scripts/e2e_cold.sh "1 64" [bytes = 30000] [out = gpurun_out/e2e_cold.txt]
cd "$(dirname "$0")/.."
. scripts/_paths.sh
LIST=${1:-"1 64"}; N=${2:-30000}; OUT=${3:-gpurun_out/e2e_cold.txt}
mkdir -p $(dirname $OUT)
W=$(mktemp -d)
{
echo "host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2), $(nproc) cpus visible, cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $N bytes per file"
for S in $LIST; do
  rm -rf $W/f; mkdir -p $W/f
  for i in $(seq 0 $((S-1))); do python3 scripts/corpus.py $W/f/$(printf %03d $i) $N $((i*157000 % 9000000)) > /dev/null; done
  s=$(date +%s.%N)
  GMX_POOL_TRACE=${TRACE:-} $(gmxbin gmix_chain_many) -T ${CHUNK:-2048} $W/out $W/f/* > $W/j.json 2> $W/err
  e=$(date +%s.%N)
  grep "gmx many" $W/err; [ -s $W/j.json ] || { echo "gmix_chain_many failed:"; tail -5 $W/err; continue; }
  python3 - $W/j.json $s $e $S $N <<'PY'
import json, sys
j = json.load(open(sys.argv[1])); s, e, S, N = float(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
bits = 8.0 * S * N
print(f"chain S={S}: process {e - s:.2f} s = {bits / (e - s):.3g} bits/s cold | in-process {j['total_seconds']:.2f} s: setup {j['build_seconds']:.2f} "
      f"(first Predictor {j['first_predictor_seconds']:.2f}, side by side: {j['parallel_construction']}), coding {j['wall_seconds']:.2f} "
      f"= {j['bits_per_second']:.3g} bits/s, teardown {j['teardown_seconds']:.2f}; failed {j['failed']}")
PY
  n=$(( S < 16 ? S : 16 ))
  s=$(date +%s.%N)
  for i in $(seq 0 $((n-1))); do ( mkdir -p $W/c$i; cd $W/c$i; $(gmxbin gmix_fast) -c $W/f/$(printf %03d $i) out > /dev/null 2>&1 ) & done; wait
  e=$(date +%s.%N)
  echo "reference CLI (-Ofast), $n processes at once: $(echo "$e $s $N $n" | awk '{printf "%.2f s = %.3g bits/s", $1-$2, 8*$3*$4/($1-$2)}')"
  same=0
  for i in 0 $((S/2)) $((S-1)); do ( mkdir -p $W/s$i; cd $W/s$i; $(gmxbin gmix_strict) -c $W/f/$(printf %03d $i) out > /dev/null 2>&1 ) & done; wait
  for i in 0 $((S/2)) $((S-1)); do cmp -s $W/s$i/out $W/out/$i.gmix && same=$((same+1)); done
  echo "identical to gmix_strict -c: $same of 3 compared"
  rm -rf $W/c* $W/s* $W/out
done
} 2>&1 | tee $OUT
rm -rf $W
