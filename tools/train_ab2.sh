# A/B of one module switch of efficient_tts_amd.train on the graphed step: train_ab2.sh NAME v0 v1 ...   (PREC=bf16|bf16x3)
name=$1; shift
for rep in 1 2 3; do
for v in "$@"; do
r=$(python tools/gpu_probe_train_graph.py ${PREC:-bf16} $name=$v 2>&1 | grep "graph" | tail -1 | grep -o "[0-9.]* ms/step")
echo "$name=$v $r"
done; done | sort | awk '{k=$1; if(!(k in m)||$2<m[k])m[k]=$2; a[k]=a[k]" "$2} END{for(k in m)print k, "min", m[k], "all", a[k]}' | sort
