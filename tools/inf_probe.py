import sys, os, json, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.getcwd())
from tools.run_configs import infer_rate
for args in [("S", 4, 10, 3, 13, 182, torch.float16), ("S", 4, 10, 3, 13, 182, torch.bfloat16), ("S", 2, 10, 3, 13, 182, torch.float32),
             ("M", 2, 10, 3, 16, 256, torch.float16), ("XS", 8, 10, 3, 4, 182, torch.float16), ("L", 1, 10, 3, 16, 356, torch.float16)]:
    try:
        print(json.dumps(infer_rate(*args)), flush=True)
    except Exception as e:
        print("ERR", args, e, flush=True)
