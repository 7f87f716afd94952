cd /root/repo
W=$(mktemp -d); cat DESIGN.md SURVEY.md > $W/c
k=0; for n in 1 300 2000 2000 777; do tail -c +$((k*5000+1)) $W/c | head -c $n > $W/f$k; k=$((k+1)); done
for rep in 1 2 3; do for exe in ${EXES:-gmix_many}; do for T in 8 1000; do
rm -rf $W/out; oracle/_ref/$exe -T $T $W/out $W/f0 $W/f1 $W/f2 $W/f3 $W/f4 > $W/j.json 2>$W/err || { echo "$exe T=$T failed"; tail -3 $W/err; }
for k in 0 1 2 3 4; do oracle/_ref/gmix_strict -c $W/f$k $W/ref$k > /dev/null 2>&1; cmp $W/ref$k $W/out/$k.gmix > $W/cmp.txt 2>&1 && echo "$exe T=$T file $k same" || echo "$exe T=$T file $k DIFFERS: $(cat $W/cmp.txt) sizes $(stat -c %s $W/ref$k) $(stat -c %s $W/out/$k.gmix)"; done
done; done; done
rm -rf $W
