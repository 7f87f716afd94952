#!/bin/bash
# The reference's own compressor end to end (`gmix -c`), stock and with its mixers (and LSTM + Indirect
# models) on the MI355X: wall time and bits/s on the same text, outputs compared.  The binaries are the
# reference built by oracle/ref_build (test infrastructure); the text is this repository's own markdown, or the
# first bytes of $GMX_CORPUS (BASELINE.json configs[0]: GMX_CORPUS=/path/to/enwik8 bash scripts/whole_pipeline_timing.sh 1000000).
#   bash scripts/whole_pipeline_timing.sh [bytes=100000]
cd "${GRAFT_REPO_ROOT:-.}"
. scripts/_paths.sh
N=${1:-100000}
W=$(mktemp -d)
if [ -n "$GMX_CORPUS" ]; then head -c $N "$GMX_CORPUS" > $W/in; else cat SURVEY.md DESIGN.md INTEGRATION.md PAPERS.md SNIPPETS.md 2>/dev/null | head -c $N > $W/in; fi
echo "input: $(wc -c < $W/in) bytes of ${GMX_CORPUS:-this repository's documents}; host: $(grep -m1 'model name' /proc/cpuinfo | cut -d: -f2)"
for exe in gmix_strict gmix_gpu gmix_chain gmix_batched gmix_chain_batched; do
  IN=$W/in
  mkdir -p $W/$exe && cd $W/$exe
  s=$(date +%s.%N)
  timeout -k 10 900 $(gmxbin $exe) -c $IN $W/$exe/out > /dev/null 2>&1
  rc=$?
  e=$(date +%s.%N)
  cd $OLDPWD
  python3 -c "import sys; n=$(wc -c < $IN); t=$e-$s; print('%-12s rc %d  %7d bytes -> %6d  %.1f s  %.0f bits/s  %.1f us/bit  md5 %s' % ('$exe', $rc, n, $(wc -c < $W/$exe/out), t, 8*n/t, t/(8*n)*1e6, '$(md5sum < $W/$exe/out | cut -c1-12)'))"
done
cmp $W/gmix_strict/out $W/gmix_gpu/out && echo "gmix_gpu output == gmix_strict output"
cmp $W/gmix_strict/out $W/gmix_chain/out && echo "gmix_chain output == gmix_strict output"
cmp $W/gmix_strict/out $W/gmix_batched/out && echo "gmix_batched output == gmix_strict output"
cmp $W/gmix_strict/out $W/gmix_chain_batched/out && echo "gmix_chain_batched output == gmix_strict output"
# and back: the stock build decodes what the device chain encoded, the device chain what the stock build encoded
(cd $W/gmix_strict && timeout -k 10 900 $(gmxbin gmix_strict) -d $W/gmix_chain/out $W/back_s > /dev/null 2>&1)
cmp $W/in $W/back_s && echo "gmix_strict -d (gmix_chain -c (input)) == input"
(cd $W/gmix_chain && timeout -k 10 900 $(gmxbin gmix_chain) -d $W/gmix_strict/out $W/back_c > /dev/null 2>&1)
cmp $W/in $W/back_c && echo "gmix_chain -d (gmix_strict -c (input)) == input"
rm -rf $W
