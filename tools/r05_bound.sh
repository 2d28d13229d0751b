# upper bounds: the graphed B=32 step with groups of launches turned into no-ops (tools/gpu_probe_train_graph.py SKIP=...)
for rep in 1 2; do
for sk in none efts_act_bwd_dropout,efts_act_bwd efts_pack_vt efts_wgrad_reduce efts_wgrad_reduce_grouped efts_wgrad_tn_grouped efts_adam_amsgrad_dev efts_layernorm_bwd efts_pack_weights_grouped efts_alpha_bwd,efts_e_bwd,efts_imv_bwd,efts_attn_bwd efts_embed_bwd efts_frame_linear efts_expand efts_sumsq; do
if [ $sk = none ]; then a=""; else a="SKIP=$sk"; fi
r=$(timeout 300 python tools/gpu_probe_train_graph.py ${PREC:-bf16} $a 2>&1 | grep "graph" | tail -1 | grep -o "[0-9.]* ms/step")
echo "BOUND $sk : $r"
done; done
