"""Static execution plan: the host-side runtime that replaces the reference's per-layer Python dispatch.

The reference walks `self.model` in Python on every forward (models/yolo_test.py:136-163) and lets ATen allocate
each intermediate.  Here a forward for a given (batch, height, width, dtype) is compiled once into a flat list of
kernel launches over pre-allocated NHWC buffers (concats resolved to channel slices at build time), which is then
either replayed launch by launch or captured into a hipGraph (icaf_graph_*) and replayed with one call.
"""
import torch

from . import ops


class ImageIn:
    """Marks a raw NCHW fp32 network input (the only non-NHWC tensor on the path)."""

    def __init__(self, tensor, c0=0, pair=None):
        """fp32: (B, 3, H, W), or both streams stacked as (2, B, 3, H, W).  uint8: the dataloader's (B, Ctot, H, W)
        batch, of which this input is channels [c0, c0+3) (pair=True: [c0, c0+3) and [c0+3, c0+6) for the two streams)."""
        self.t, self.c0 = tensor, c0
        self.u8 = tensor.dtype == torch.uint8
        assert tensor.dim() == 4 if self.u8 else tensor.dim() in (4, 5)
        self.pair = bool(pair) if self.u8 else tensor.dim() == 5

    @property
    def shape(self):
        B, _, H, W = self.t.shape[-4:]
        return (B, 3, H, W)


class Plan:
    def __init__(self, device, dtype):
        self.device, self.dtype = torch.device(device), dtype
        self.launches = []
        self.inputs = []        # static input tensors, filled by the caller before run()
        self.outputs = None
        self.graph = None
        self.nbytes = 0
        # side branches of the captured graph: id -> {"after": index of the launch the branch depends on,
        # "join_before": index of the first main-chain launch that needs its results}.  Launches tagged with the id run
        # on their own capture stream, i.e. as a parallel branch of the hipGraph: small latency-bound kernels (DMFF of
        # the shallow levels) fill the CUs the backbone's tails leave idle.  Eager replay ignores branches.
        self.branches = {}
        self.notes = {}         # plan-build facts that launch names do not show (fusion_report reads them)

    # -- buffers ------------------------------------------------------------------------------------------
    def act(self, B, H, W, C, dtype=None, pair=False):
        """NHWC activation buffer; pair=True allocates both backbone streams adjacently as (2, B, H, W, C)."""
        t = torch.zeros(((2, B, H, W, C) if pair else (B, H, W, C)), dtype=dtype or self.dtype, device=self.device)
        self.nbytes += t.numel() * t.element_size()
        return t

    def tokens(self, G, rows, C, dtype=None):
        t = torch.zeros((G, rows, C), dtype=dtype or self.dtype, device=self.device)
        self.nbytes += t.numel() * t.element_size()
        return t

    def empty(self, shape, dtype):
        t = torch.zeros(shape, dtype=dtype, device=self.device)
        self.nbytes += t.numel() * t.element_size()
        return t

    def add(self, launch):
        self.launches.append(launch)
        return launch

    # -- execution ----------------------------------------------------------------------------------------
    def run(self, stream_ptr=None):
        sp = stream_ptr if stream_ptr is not None else ops.current_stream_ptr()
        if self.graph is not None:
            self.graph.launch(sp)
        else:
            for l in self.launches:
                l(sp)

    def autotune(self, stream_ptr=None):
        """Pick the fastest igemm configuration for every conv / linear launch of this plan (device-timed)."""
        sp = stream_ptr if stream_ptr is not None else ops.current_stream_ptr()
        fn = ops.lib().icaf_conv2d
        for l in self.launches:             # one untimed pass first: lazy module load, attribute setup
            l(sp)
        # in situ: each candidate is timed right after a replay of the (up to) 6 launches in front of the layer
        return [ops.autotune_conv(l, sp, context=self.launches[max(0, i - 6):i])
                for i, l in enumerate(self.launches) if l.fn is fn]

    def capture(self):
        """Capture the launch list into a hipGraph on side streams (the legacy default stream cannot capture); launches
        of a branch are captured on their own stream between a fork and a join event."""
        dev = self.device
        main = torch.cuda.Stream(device=dev)
        main.wait_stream(torch.cuda.current_stream(dev))
        sp = main.cuda_stream
        for l in self.launches:          # warm every kernel once outside capture (module load, first-touch)
            l(sp)
        main.synchronize()
        sides = {bid: torch.cuda.Stream(device=dev) for bid in self.branches}
        forks = {bid: ops.Event() for bid in self.branches}
        joins = {bid: ops.Event() for bid in self.branches}
        self._capture_keep = (main, sides, forks, joins)

        def body():
            started, joined = set(), set()

            def join(bid):
                joins[bid].record(sides[bid].cuda_stream)
                joins[bid].wait(sp)
                joined.add(bid)
            for idx, l in enumerate(self.launches):
                for bid, b in self.branches.items():
                    if b["join_before"] == idx and bid in started and bid not in joined:
                        join(bid)
                if l.branch:
                    bid = l.branch
                    if bid not in started:
                        forks[bid].wait(sides[bid].cuda_stream)
                        started.add(bid)
                    l(sides[bid].cuda_stream)
                else:
                    l(sp)
                for bid, b in self.branches.items():
                    if b["after"] == idx:
                        forks[bid].record(sp)
            for bid in started - joined:
                join(bid)
        g = ops.Graph()
        g.capture(sp, body)
        main.synchronize()
        self.graph = g
        return self

    def fusion_report(self):
        """Which of the width-specialised fused launches this plan runs, read off its launch names — so that a model whose widths fall outside a
        specialisation (icaf_stem2: 3 -> 32 -> 64 -> 2 x 32; icaf_bottleneck: c_ in {32, 64}; the three-launch DMFF block: C in {128, 256, 512})
        shows it instead of silently running the slower per-layer launches.  bench.py prints it as config.fused_paths."""
        names = [l.name for l in self.launches]
        n = names.count
        dmff = ("three-launch" if n("dmff_proj_mlp") else "") + ("+two-launch" if n("dmff_attn_mlp") else "") + ("+per-layer" if n("mlp_fc1") else "")
        return {"launches": len(names), "stem": "stem2 (stem + 3x3/s2 + 1x1)" if n("stem+conv3x3s2+1x1") else "stem" if n("stem") else "staging + conv",
                "bottleneck_fused": sum(1 for x in names if x.startswith("bottleneck")), "c3_tails": sum(1 for x in names if x.endswith("+cv3")),
                "chained_1x1": sum(1 for x in names if x.endswith("+1x1") and not x.startswith("stem")),
                "dmff_blocks": dmff.strip("+") or "none", "dmff_levels_three_launch": n("dmff_proj_mlp"), "dmff_levels_per_layer": n("mlp_fc1"),
                "detect": "conv+decode fused" if n("detect_conv+decode") else "conv, decode",
                # loops > 1: which DMFF levels (by C) carry the residual token stream in fp32 from one iteration to the next, and which round it to
                # the storage type twice per iteration (not built: the two-launch form, C = 512 without a hidden split, the per-layer form) — a mixed
                # configuration is visible here instead of in a parity margin
                "dmff_fp32_token_stream": self.notes.get("dmff_fp32_token_stream")}

    def timed_run(self, stream_ptr=None):
        """Run launch by launch with a HIP event pair around every kernel; returns [(name, ms, flops, bytes)]."""
        sp = stream_ptr if stream_ptr is not None else ops.current_stream_ptr()
        evs = [ops.Event() for _ in range(len(self.launches) + 1)]
        evs[0].record(sp)
        for i, l in enumerate(self.launches):
            l(sp)
            evs[i + 1].record(sp)
        out = []
        for i, l in enumerate(self.launches):
            out.append((l.name, evs[i].elapsed_ms(evs[i + 1]), l.flops, l.bytes))
        return out


def concat_view(xs):
    """If the acts are adjacent channel slices of one buffer (in order), return the covering view, else None."""
    first = xs[0]
    es = first.element_size()
    ptr = first.data_ptr()
    total = 0
    for t in xs:
        if t.data_ptr() != ptr + total * es or t.stride() != first.stride() or t.shape[:3] != first.shape[:3]:
            return None
        total += t.shape[3]
    if total > first.stride(2):
        return None
    B, H, W, _ = first.shape
    return first.as_strided((B, H, W, total), first.stride())


def to_act(x, dtype):
    """NCHW torch tensor -> NHWC act in dtype (boundary plumbing for stand-alone module calls)."""
    return x.permute(0, 2, 3, 1).contiguous().to(dtype)


def from_act(y):
    """NHWC act -> NCHW-shaped tensor (channels_last memory, no copy)."""
    return y.permute(0, 3, 1, 2)
