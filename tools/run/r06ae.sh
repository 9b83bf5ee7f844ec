# round 6: one more stress campaign on the product build at the last code state (new seeds)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06ae; mkdir -p $O
run() { name=$1; shift; timeout 900 python tools/stress_lap.py "$@" > $O/$name.log 2>&1; echo "$name ($*): $(tail -1 $O/$name.log)" | tee -a $O/summary.txt; }
rm -f $O/summary.txt
run a 5000 240 100 3000
run b 5300 60 3000 7000
run c 5400 90 200 3000 --par 7 --wipe 1
run d 5500 90 200 3000 --rebuild -1
run e 5600 60 200 3000 --groups 4
run f 5700 60 200 2500 --chain
run g 5800 40 200 2500 --f64
cat $O/summary.txt
