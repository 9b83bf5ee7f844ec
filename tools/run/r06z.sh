# round 6: randomised stress at the round's last code state (product build; and the -DCYTO_COOP_MIN_N=0 build for the cooperative full-row path)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06z; mkdir -p $O
C0=$GRAFT_REPO_ROOT/cytospace_amd/build/libcytohip_coop0.so
run() { name=$1; shift; timeout 1100 python tools/stress_lap.py "$@" > $O/$name.log 2>&1; echo "$name ($*): $(tail -1 $O/$name.log)" | tee -a $O/summary.txt; }
rm -f $O/summary.txt
run a 3000 150 100 3000
run b 3100 40 3000 7000
run c 3200 60 200 3000 --par 4 --wipe 1
run d 3300 60 200 3000 --wipe 2 --rounds 30
run e 3400 60 200 3000 --rebuild 2
run f 3500 40 200 2500 --chain
run g 3600 30 200 2500 --f64
CYTOHIP_LIB=$C0 run h_coop0 3700 80 200 4000
CYTOHIP_LIB=$C0 run i_coop0 3800 40 500 3000 --par 5
CYTOHIP_LIB=$C0 run j_coop0 3900 40 500 3000 --rebuild -1
cat $O/summary.txt
