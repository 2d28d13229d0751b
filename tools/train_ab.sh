for rep in 1 2 3; do
for cfg in ${CFGS:-"1 0" "1 1" "3 0" "3 1" "3 3"}; do
set -- $cfg
r=$(python tools/gpu_probe_train_graph.py ${PREC:-bf16} _RESCONV_FWD=$1 _RESCONV_DGRAD=$2 2>&1 | grep "graph" | tail -1 | grep -o "[0-9.]* ms/step")
echo "FWD=$1 DGRAD=$2 $r"
done; done | sort | awk '{k=$1" "$2; if(!(k in m)||$3<m[k])m[k]=$3; a[k]=a[k]" "$3} END{for(k in m)print k, "min", m[k], "all", a[k]}' | sort
