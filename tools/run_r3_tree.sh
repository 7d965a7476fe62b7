O=gpurun_out/r3_topk_tree.txt; : > $O
for rep in 1 2; do for dt in fp32 fp16; do for v in notree tree; do DT=$dt tools/micro/rank_forward_lat_$v 2>&1 | grep -v "^fp" | sed "s/^/$dt $v /" | tee -a $O; done; done; done
