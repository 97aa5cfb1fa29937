"""TEST ONLY stand-in for the HIP engine, selected by GTCRN_BENCH_TEST_SHIM (tests/test_dist_gloo.py): lets
bench.py's multi-rank control flow (self-launch, rendezvous, barriers, max over ranks, one JSON line) run on
CPU over gloo.  The numbers bench.py prints with it are meaningless and labelled as such."""


class _Engine:
    def __init__(self, steps):
        self.steps = steps

    def forward_wave(self, w, win, out=None):
        out.copy_(w[:, :out.shape[1]] * 0.5)

    def timing_enable(self, on=True, only=None):
        pass

    def timing_read(self):
        return {k: (1.0, self.steps) for k in ("k_stft", "k_encoder", "k_gtcn1", "k_gtcn2", "k_decoder", "k_istft")}

    def reserve(self, B, T):
        pass


def make_engine(params, local_rank, args):
    return _Engine(args.steps)


def secondary_legs(rank, world):
    """TEST ONLY (GTCRN_BENCH_TEST_LEGS): stand-in secondary legs for bench.py's run_leg / Watchdog machinery.
    "fail1": rank 1 raises inside its leg; "hang1": rank 1 never comes back from its leg."""
    import os
    import time
    mode = os.environ.get("GTCRN_BENCH_TEST_LEGS", "")
    if not mode:
        return {}

    def ok_leg(sync_local, record):
        sync_local()
        record(0.010 * (1 + rank))                     # rank 1 is the slower one: it sets the aggregate rate
        return {"units_per_s": 1000.0, "ms": 10.0, "_rate_keys": ["units_per_s"], "_time_keys": ["ms"]}

    def bad_leg(sync_local, record):
        if rank == 1 and mode == "fail1":
            raise RuntimeError("simulated out-of-memory on rank 1")
        if rank == 1 and mode == "hang1":
            time.sleep(3600)
        record(0.010)
        return {"units_per_s": 1.0, "_rate_keys": ["units_per_s"]}

    return {"first": (ok_leg, 60.0), "second": (bad_leg, 6.0 if mode == "hang1" else 60.0), "third": (ok_leg, 60.0)}
