for mode in lazy single; do
  for args in "1000 4000 10000" "20000 27000" "--dup=5 5000 10000 20000"; do
    echo "mode=$mode $args"
    CYTO_AUG=$mode timeout 300 python tools/quick_lap_bench.py $args 2>&1 | grep -o "n=[0-9]* \|augrelax=[0-9]*\|arr_ms=[0-9.]* aug_ms=[0-9.]* us/arr=[0-9.]* us/aug=[0-9.]*\|aug_skipped=[0-9]*\|aug_dense=[0-9]*" | paste - - - - -
  done
done
