#!/usr/bin/env python3
"""Entry point with the reference's name and flags (pretrain_eval_attention.py --exp_path ... --out_path ... --data_type ...):
the evaluation sweep of 6dgs_amd/pretrain_eval_attention.py.  One process, or one per GPU under torch.distributed.run:

    python pretrain_eval_attention.py --exp_path output --out_path results/pose_eval.json --data_type mip360
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 pretrain_eval_attention.py --exp_path output --out_path r.json
"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

if __name__ == "__main__":
    if os.environ.get("SIXDGS_FORCE_DEVICE") is not None and int(os.environ.get("WORLD_SIZE", "1")) > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "1")      # test hook, several ranks on ONE GPU: see bench.py main() / profiles/r06_eight_ranks_sigabrt.md
    import torch
    torch.manual_seed(71170)            # the reference seeds here too (pretrain_eval_attention.py:252-253)
    importlib.import_module("6dgs_amd.pretrain_eval_attention").main()
