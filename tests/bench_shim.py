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
