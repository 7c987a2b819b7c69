#!/bin/bash
run() { n=$1; shift; env "$@" timeout -k 10 280 python bench.py --steps 2 --warmup 1 --no-cpu 2>gpurun_out/var_$n.err > gpurun_out/var_$n.json; python -c "
import json,sys; b=json.load(open('gpurun_out/var_$n.json')); k=b['kernel_ms']; print('$n', 'paths', round(k['paths'],1), 'mid', round(k['mid_build'],1), 'rev', round(k['reverse'],1), 'knn', round(k['knn_classify'],1), 'B', round(b['stage_ms']['B_extend'],1), 'A', round(b['stage_ms']['A_item_sim'],2), 'C', round(b['stage_ms']['C_generate'],2), 'prof/s', round(b['alterego_profiles_per_s']))"; }
