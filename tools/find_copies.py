"""Which host-side operation issues the device-to-device copies of a train step?  (202 __amd_rocclr_copyBuffer
launches per step in profiles/r05_train_f32_kernel_stats.csv.)  Runs a few train steps at a small batch under
torch.profiler and prints, per CPU-side op, how many Memcpy DtoD / copy kernels it launched.

    python tools/find_copies.py [--batch 32] [--storage f32]
"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--storage", default="f32")
    a = ap.parse_args()
    import torch
    from torch.profiler import profile, ProfilerActivity
    from gtcrn_micro_amd.train import make_training, synthetic_mix, train_step
    torch.manual_seed(43)
    model, opt, sched, loss_func = make_training(device="cuda")
    model.train()
    if a.storage != "f32":
        model.set_activation_storage(a.storage)
    noisy, clean = synthetic_mix(a.batch, samples=64000, seed=43)
    for _ in range(2):
        train_step(model, opt, sched, loss_func, noisy, clean)
    torch.cuda.synchronize()
    steps = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for _ in range(steps):
            train_step(model, opt, sched, loss_func, noisy, clean)
        torch.cuda.synchronize()
    evs = prof.events()
    dev = [e for e in evs if e.device_type == torch.autograd.DeviceType.CUDA]
    names = collections.Counter(e.name for e in dev)
    print(f"device activities over {steps} steps (top 25 by count):")
    for n, c in names.most_common(25):
        print(f"  {c:6d}  {n[:110]}")
    # CPU ops that launched a memcpy: walk the CPU events, count their direct kernels
    by_op = collections.Counter()
    stacks = {}
    for e in evs:
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        for k in getattr(e, "kernels", []) or []:
            if "emcpy" in k.name or "copyBuffer" in k.name:
                by_op[e.name] += 1
                if e.stack:
                    stacks.setdefault(e.name, collections.Counter())[" <- ".join(s.split("/")[-1] for s in e.stack[:4])] += 1
    print("\nCPU ops that launched copies:")
    for n, c in by_op.most_common(20):
        print(f"  {c:6d}  {n}")
        for st, cc in stacks.get(n, {}).most_common(4):
            print(f"            {cc:5d} x {st[:160]}")


if __name__ == "__main__":
    main()
