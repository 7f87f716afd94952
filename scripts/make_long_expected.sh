#!/bin/bash
# BUILD CONTAINER ONLY (needs oracle/_ref/gmix_strict, i.e. /root/reference): what the reference's strict stock build
# makes of the long inputs of scripts/e2e_long.py -- one 10^7-byte stream and 64 windows of 10^6 bytes of the standard
# library's sources (scripts/corpus.py) -- as md5 sums in tests/golden/long_expected.json.  About 35 minutes of 8 cores.
#   bash scripts/make_long_expected.sh [work dir = /tmp/long] [parallel jobs = 6]
cd "$(dirname "$0")/.."
ROOT=$PWD
W=${1:-/tmp/long}; P=${2:-6}
mkdir -p $W && python3 scripts/corpus.py $W/all.txt > $W/corpus.info
cat > $W/run_expected.sh <<EOS
#!/bin/bash
set -e
name=\$1; n=\$2; off=\$3
d=$W/job_\$name
mkdir -p \$d && cd \$d
python3 -c "import sys; d=open('$W/all.txt','rb').read(); open('in','wb').write(d[\$off:\$off+\$n])"
s=\$(date +%s.%N)
$ROOT/oracle/_ref/gmix_strict -c in out > log 2>/dev/null
e=\$(date +%s.%N)
echo "\$name \$n \$off \$(md5sum < in | cut -d' ' -f1) \$(stat -c %s out) \$(md5sum < out | cut -d' ' -f1) \$(python3 -c "print(\$e - \$s)")" >> $W/expected.txt
rm -rf analysis
EOS
chmod +x $W/run_expected.sh
[ -s $W/expected.txt ] || { ( echo "big 10000000 0"; for k in $(seq 0 63); do echo "f$k 1000000 $((k*157000))"; done ) | nice -n 5 xargs -P $P -L 1 $W/run_expected.sh; }
python3 - $W <<'PY'
import hashlib, json, sys
w = sys.argv[1]
rows = {r[0]: r for r in (l.split() for l in open(w + "/expected.txt"))}
data = open(w + "/all.txt", "rb").read()
def rec(r): return {"bytes": int(r[1]), "offset": int(r[2]), "in_md5": r[3], "out_bytes": int(r[4]), "out_md5": r[5], "strict_seconds_in_build_container": round(float(r[6]), 1)}
out = {"what": "gmix_strict -c (the reference's CLI, g++ -O2 strict build: oracle/ref_build/Makefile) on windows of the corpus of scripts/corpus.py",
       "corpus_bytes": len(data), "corpus_md5": hashlib.md5(data).hexdigest(), "big": rec(rows["big"]),
       "files": [rec(rows[f"f{k}"]) for k in range(64)]}
json.dump(out, open("tests/golden/long_expected.json", "w"), indent=1)
print("tests/golden/long_expected.json:", len(out["files"]), "files +", out["big"]["bytes"], "bytes")
PY
