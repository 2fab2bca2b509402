#!/usr/bin/env python
"""Copies the summaries of a tools/profile.sh run (gpurun_out/prof_rNN) into profiles/rNN (tracked).
    python tools/collect_profiles.py gpurun_out/prof_r02a profiles/r02"""
import os
import shutil
import sys

src, dst = sys.argv[1], sys.argv[2]
os.makedirs(dst, exist_ok=True)
pairs = [("kernel_stats.csv", "kernel_stats.csv"), ("pmc_summary.csv", "pmc_summary.csv"), ("bench.json", "bench.json"),
         ("bench_under_rocprof.json", "bench_under_rocprof.json"), ("fwd_split/kernel_stats.csv", "forward_split_kernel_stats.csv"),
         ("fwd_split/pmc_summary.csv", "forward_split_pmc_summary.csv"), ("fwd_f32/kernel_stats.csv", "forward_f32_kernel_stats.csv"),
         ("fwd_f32/pmc_summary.csv", "forward_f32_pmc_summary.csv"), ("fwd_bf16/kernel_stats.csv", "forward_bf16_kernel_stats.csv"),
         ("fwd_bf16/pmc_summary.csv", "forward_bf16_pmc_summary.csv"), ("train/kernel_stats.csv", "train_kernel_stats.csv"),
         ("power_trace_b24.txt", "power_trace_b24.txt"), ("power_trace_bf16_512_b8.txt", "power_trace_bf16_512_b8.txt"),
         ("fwd_graph_ab.txt", "fwd_graph_ab.txt"), ("bf16_layer_table.txt", "bf16_layer_table.txt"), ("map_bins_ab.txt", "map_bins_ab.txt"), ("train_trace.txt", "train_trace.txt"), ("train_pmc_summary.txt", "train_pmc_summary.txt"),
         ("train_host_time.txt", "train_host_time.txt"), ("bench_bn.txt", "bench_bn.txt"), ("bench_train.json", "bench_train.json"),
         ("forward_layers_b24.txt", "forward_layers_b24.txt"), ("forward_layers_b1.txt", "forward_layers_b1.txt")]
for a, b in pairs:
    p = os.path.join(src, a)
    if os.path.exists(p) and os.path.getsize(p) > 0:
        shutil.copy(p, os.path.join(dst, b))
        print("copied", a, "->", b)
    else:
        print("missing", a)
