"""Is the 2-rank training deterministic run to run, with and without the overlapped table-gradient all-reduce?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_ddp as T

if __name__ == "__main__":
    stage = sys.argv[1] if len(sys.argv) > 1 else "instance"
    runs = []
    for ov in (False, False, False, True, True, True):
        runs.append((ov, T._run(stage, ov)[0][2]["instance_encoder.embeddings" if stage == "instance" else "encoder.embeddings"]))
    for i in range(len(runs)):
        for j in range(i + 1, len(runs)):
            a, b = runs[i][1], runs[j][1]
            print(runs[i][0], runs[j][0], f"{float(np.mean(np.abs(a - b) > 1e-4 + 1e-3 * np.abs(b))):.2e}", f"max {np.abs(a-b).max():.2e}")
