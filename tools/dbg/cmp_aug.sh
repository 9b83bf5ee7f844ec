for mode in lazy single; do
  for args in "2000 4000 6000 8000 10000 12000" "--dup=5 2000 4000 6000 8000 10000" "--dup=10 5000 10000"; do
    echo "mode=$mode $args"
    CYTO_AUG=$mode timeout 300 python tools/quick_lap_bench.py $args 2>&1 | grep -o "n=[0-9]* \|augrelax=[0-9]*\|aug_ms=[0-9.]* \|us/aug=[0-9.]*" | paste - - - -
  done
done
