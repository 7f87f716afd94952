cd "$(dirname "$0")/.."
W=$(mktemp -d); cat DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md DESIGN.md SURVEY.md INTEGRATION.md > $W/c
for S in ${1:-96 128}; do
  rm -rf $W/f; mkdir $W/f; for i in $(seq 0 $((S-1))); do tail -c +$((i*1531+1)) $W/c | head -c ${2:-30000} > $W/f/$i; done
  ${PFX:-} dropin/_build/gmix_chain_many -T ${CHUNK:-2048} $W/out $W/f/* | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('chain S=%d: %.3g bits/s, %.2f us/bit/stream, wall %.2f submit %.2f wait %.2f launches %d' % (j['files'], j['bits_per_second'], j['wall_seconds']*1e6/(8*j['jobs'][0]['in']), j['wall_seconds'], j['submit_seconds'], j['wait_seconds'], j['launches']))"
done
rm -rf $W
