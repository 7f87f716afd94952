cd /root/repo
W=$(mktemp -d); cat DESIGN.md SURVEY.md > $W/c
k=0; for n in 1 300 2000 2000 777; do tail -c +$((k*5000+1)) $W/c | head -c $n > $W/f$k; oracle/_ref/gmix_strict -c $W/f$k $W/ref$k > /dev/null 2>&1; k=$((k+1)); done
chk() { for k in 0 1 2 3 4; do [ -f $W/out/$k.gmix ] && { cmp -s $W/ref$k $W/out/$k.gmix && printf " ok" || printf " BAD$k"; }; done; echo; }
echo "alone:"; for k in 0 1 2 3 4; do rm -rf $W/out; oracle/_ref/gmix_many -T 1000 $W/out $W/f$k > /dev/null 2>&1; cmp -s $W/ref$k $W/out/0.gmix && printf " ok" || printf " BAD$k"; done; echo
echo "equal lengths (4 x file2):"; rm -rf $W/out; oracle/_ref/gmix_many -T 1000 $W/out $W/f2 $W/f2 $W/f2 $W/f2 > /dev/null 2>&1; for k in 0 1 2 3; do cmp -s $W/ref2 $W/out/$k.gmix && printf " ok" || printf " BAD$k"; done; echo
echo "two files (2000, 777):"; rm -rf $W/out; oracle/_ref/gmix_many -T 1000 $W/out $W/f2 $W/f4 > /dev/null 2>&1; cmp -s $W/ref2 $W/out/0.gmix && printf " ok" || printf " BAD"; cmp -s $W/ref4 $W/out/1.gmix && echo " ok" || echo " BAD"
echo "five:"; for r in 1 2 3; do rm -rf $W/out; oracle/_ref/gmix_many -T 1000 $W/out $W/f0 $W/f1 $W/f2 $W/f3 $W/f4 > /dev/null 2>&1; chk; done
echo "five, no pin:"; for r in 1 2; do rm -rf $W/out; oracle/_ref/gmix_many -T 1000 --no-pin $W/out $W/f0 $W/f1 $W/f2 $W/f3 $W/f4 > /dev/null 2>&1; chk; done
rm -rf $W
