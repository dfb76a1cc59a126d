"""Two-stage detection pipeline: forward on one HIP stream, NMS (+ the detections all-gather) on a second one.

The reference runs `model(img, img2)` and `non_max_suppression` back to back for every batch with a device sync in
between (detect_twostream.py:83-88, test.py:126-141).  NMS is a latency-bound job that occupies one workgroup per
image (32 of 256 CUs at batch 32), so here batch i's NMS runs concurrently with batch i+1's forward:

    forward stream : graph(i) -> snapshot z(i) -> graph(i+1) -> snapshot z(i+1) -> ...
    nms stream     :              wait snapshot(i) -> candidates / sort / greedy (i) -> ...
    gather stream  :  (N > 1)                 wait the NMS of the group's last batch -> ONE all_gather of the group's detection blocks

With `depth` > 1 (bench default 2) that many batches are in flight: batch n replays plan n % nplans (own buffers, own hipGraph) on
forward stream n % depth, so the low-occupancy tail of one forward overlaps the full-width layers of the next (+13 % throughput on one
MI355X); NMS then reads each plan's own `z` and the plan is not replayed before its NMS has finished.  nplans = depth,
or a multiple of it when the pipeline is fed from host memory (u8=True): the host -> device copy of a batch then lands DIRECTLY in the
input buffer of a plan that no forward in flight is reading — one PCIe copy per batch and no device-to-device hop.

With depth 1, `z` is snapshotted into one of two staging buffers because the plan's output buffer is overwritten by the next
replay; events order snapshot -> NMS -> reuse.  Each of the two slots also owns its NMS runner (workspace + det / count /
keep output buffers) and its gathered block, so the tensors step n returned stay untouched until step n + 2 reuses the
slot — and two pipelines of the same shape never share buffers.  Results of step i are valid after `synchronize()`.
PyTorch streams / events are used as plumbing only; every kernel on both streams is ours (plus RCCL).
"""
import torch

from . import dist as D
from . import ops
from .options import OPT
from .utils.general import nms_device


# further SETS of `depth` plans a host-fed pipeline owns: the copy of batch n waits for the END of the forward that used its target plan,
# depth * (1 + EXTRA_PLANS) steps earlier.  With chain graphs (see HOST_FED_BRANCHES) one extra set is enough — one process each, same box:
# 4 plans 14,745 / 14,983 pairs/s, 6 plans 14,605 (and 6 plans pin 11 GB for yolov5s batch 32); with branched graphs it took 6 plans to reach
# 13,400 (4 plans: 10,742)
# (Memory: a host-fed pipeline therefore holds depth * (1 + EXTRA_PLANS) COMPLETE plans — intermediates, hipGraph, NMS runner each: 2.0 GB per plan for
#  yolov5s batch 32 at 640 x 640, ~40 GB for yolov5l batch 16 at 1280 x 1280, i.e. 160 GB for the four plans of a depth-2 yolov5l VEDAI pipeline.  That
#  fits the 288 GB of an MI355X and nothing smaller; Model.plan_cache_bytes cannot evict plans a pipeline holds.)
EXTRA_PLANS = max(1, OPT.pipe_extra_plans)
COPY_STREAMS = max(1, OPT.pipe_copy_streams)      # a batch's host -> device copy in this many slices, one high-priority stream each
COPY_PRIO = OPT.pipe_copy_prio                    # the copy stream(s) on a high-priority queue: 15,600 / 15,501 pairs/s against 15,288 / 15,369 at normal priority
HOST_FED_BRANCHES = OPT.pipe_branches             # keep the hipGraph's parallel branches in host-fed pipelines (measured slower)
# (round 5's icaf_feed_copy — resident workgroups reading the pinned batch over PCIe instead of the DMA engine — measured 2 x slower inside the
#  serving loop and was removed in round 6)


class DetectionPipeline:
    def __init__(self, model, batch, height, width, device, conf_thres=0.25, iou_thres=0.45, classes=None,
                 agnostic=False, multi_label=False, max_det=300, world=1, overlap=True, force_gather=False, depth=1, u8=False):
        """u8=True: the plans take the dataloader's uint8 (B, 6, H, W) RGB+IR batch (Model.forward_u8: `/255`, split and cast in the
        staging kernel, reference test.py:116-123) — `submit_u8` then feeds them from (pinned) host memory with the H2D copy on its own
        stream, overlapped with the forwards in flight."""
        self.model, self.device, self.world = model, torch.device(device), world
        self.u8 = bool(u8)
        self.gather = world > 1 or bool(force_gather)       # force_gather: run the all-gather even with one rank (hardware test of the RCCL path)
        self.nms_args = dict(conf_thres=conf_thres, iou_thres=iou_thres, classes=classes, agnostic=agnostic,
                             multi_label=multi_label, max_det=max_det)
        # (the pipeline replays the plans itself and reads plan.outputs: it does not touch Model.static_outputs — a later model(rgb, ir) of the
        #  same object still returns clones)
        # depth > 1: that many batches in flight, each with its own plan (buffers, hipGraph) and forward stream — the tails of one
        # forward (20x20 layers, DMFF, Detect: launches that leave CUs idle) overlap the full-width layers of the next
        self.depth = max(1, int(depth)) if overlap else 1         # overlap=False is the strictly sequential baseline: one batch, one stream
        # host-fed pipelines own ONE MORE plan than batches in flight: the copy of the next batch goes straight into the input of the plan that
        # is not in flight (round 4 copied into depth + 1 staging buffers and moved the batch into the plan's input device-to-device: 20 % of
        # the no-feed rate was lost to that hop and to the queue it shared)
        # (a MULTIPLE of depth: plan p then always replays on forward stream p % depth — a hipGraph keeps its stream — and the pipeline owns no more
        #  forward streams than forwards in flight: HIP deals a process's streams over a few hardware queues, and every extra stream is one more
        #  chance that two of them that should overlap share one)
        self.nplans = self.depth * (1 + EXTRA_PLANS) if (self.u8 and overlap) else self.depth
        # (host-fed: chain graphs.  A graph with parallel branches replays them on streams of its own, which share hardware queues with the copy
        #  and NMS streams: same box, one process each, 13,578 pairs/s with the branches, 15,153 without, 15,172 with no graph at all — and
        #  16,100-16,500 with the inputs resident, where the branches are worth 3 %)
        self.plans = [model.plan_for(batch, height, width, self.device, u8=self.u8, slot=s, branches=HOST_FED_BRANCHES or not (self.u8 and overlap))
                      for s in range(self.nplans)]
        # (a HIGH-PRIORITY stream: HIP maps the streams of a process onto a few hardware queues, and a copy stream that shares its queue with
        #  a forward stream waits behind that stream's graph — the copies then do not overlap the forwards at all; priority streams get
        #  queues of their own)
        self.copy_streams = [torch.cuda.Stream(device=self.device, priority=COPY_PRIO) for _ in range(COPY_STREAMS)] if self.u8 else []
        self.copy_stream = self.copy_streams[0] if self.u8 else None
        self.copied = [[torch.cuda.Event() for _ in self.copy_streams] for _ in self.plans]
        self.plan = self.plans[0]
        self.z = self.plan.outputs[0]
        # `depth` forward streams; plan p always replays on stream p % depth (alternating a graph between two streams cost the host-fed loop a
        # third of its rate), so at most `depth` forwards run at once by construction
        self.fwd_streams = [torch.cuda.Stream(device=self.device) for _ in range(self.depth)]
        self.fwd_stream = self.fwd_streams[0]
        self.fwd_done = [torch.cuda.Event() for _ in range(self.nplans)]
        self.nms_stream = torch.cuda.Stream(device=self.device) if overlap else self.fwd_stream
        self.overlap = overlap
        self.zbuf = [torch.empty_like(self.z) for _ in range(2)] if overlap else [self.z]
        self.snap_done = [torch.cuda.Event() for _ in range(2)]
        self.nms_done = [torch.cuda.Event() for _ in range(2)]
        nslots = 2 if overlap else 1
        rows, no = self.z.shape[1], self.z.shape[2]
        ml = bool(multi_label) and no - 5 > 1
        # The detection blocks of the slots lie SIDE BY SIDE in one allocation: with a process group the all-gather then runs once per GROUP of
        # steps (one per in-flight slot: `depth` batches with the bench's default pipeline) on a stream of its own behind the group's last NMS —
        # one latency-bound collective of group x 230 KB instead of `group` of them, and none of them on the NMS stream, where the RCCL kernel
        # sat between two batches' NMS launches (one GPU, --force-gather: -3.5 % with a gather per step on the NMS stream).
        blk = batch * max_det * 6 + batch
        self.group = self.nplans if self.nplans > 1 else 1          # steps per collective (the one-plan pipeline alternates two slots: one step each)
        nblocks = self.nplans if self.nplans > 1 else nslots
        self.group_block = torch.zeros((nblocks * blk,), dtype=torch.float32, device=self.device)
        slot_block = [self.group_block[k * blk:(k + 1) * blk] for k in range(nblocks)]
        self.runners = [ops.NmsRunner(batch, rows, no - 5, self.device, ml, max_det, want_keep=False, block=slot_block[k] if self.nplans == 1 else None)
                        for k in range(nslots)]
        if self.nplans > 1:
            self.deep_runners = [ops.NmsRunner(batch, rows, no - 5, self.device, ml, max_det, want_keep=False, block=slot_block[k]) for k in range(self.nplans)]
            self.nms_done_deep = [torch.cuda.Event() for _ in range(self.nplans)]
        # two generations of the gathered buffer: the tensors a step returned stay untouched until the gather AFTER the next one
        self.gathered = [torch.empty((world * self.group * blk,), dtype=torch.float32, device=self.device) for _ in range(2)] if self.gather else None
        self.gather_stream = torch.cuda.Stream(device=self.device) if (self.gather and overlap) else self.nms_stream
        self.gather_done = torch.cuda.Event()
        self.nms_group_done = torch.cuda.Event()
        self.gen, self.pending, self.gathers = 0, 0, 0               # generation the NEXT gather writes; steps since the last gather; gathers issued
        self.n = 0
        self.last = None

    @property
    def inputs(self):
        """Static RGB / IR staging tensors (NCHW fp32) of the NEXT step's plan.  With depth > 1 every in-flight slot has its OWN
        staging tensors: they must be refilled before EVERY step (filling them once and stepping repeatedly would run every
        other batch on another slot's stale inputs), and the caller's copy must be ordered against the slot's forward stream —
        `submit()` does both."""
        return self.plans[self.n % self.nplans].inputs

    def submit(self, rgb, ir):
        """Copy one batch into the next step's staging tensors ON that step's forward stream, behind the plan's previous forward (the same
        stream when nplans == depth; an event otherwise) so a forward still reading them is never overwritten, then enqueue the step."""
        pi = self.n % self.nplans
        fs = self.fwd_streams[pi % self.depth]
        ins = self.plans[pi].inputs
        fs.wait_stream(torch.cuda.current_stream(self.device))     # rgb / ir may have been produced on the caller's stream
        with torch.cuda.stream(fs):                                # (the plan's own stream: behind its previous forward)
            ins[0].copy_(rgb, non_blocking=True)
            ins[1].copy_(ir, non_blocking=True)
        for t in (rgb, ir):                                        # the copies run on `fs`, possibly long after this call returns: tell the
            if t.is_cuda:                                          # caching allocator, or a temporary's memory is handed out (and overwritten
                t.record_stream(fs)                                # on the caller's stream) before the copy has read it
        return self.step()

    def submit_u8(self, img6):
        """One uint8 (B, 6, H, W) batch — pinned host memory (the reference's dataloader output, test.py:116) or a device tensor — through
        the pipeline.  The copy runs on the pipeline's COPY stream straight into the input buffer of plan n % nplans: the plan that ran
        nplans (= depth * (1 + EXTRA_PLANS)) steps ago, so the copy only waits for a forward that has long finished — NOT for one of the `depth` forwards in flight (a
        copy into the input of a plan that is still running could not start before that forward had ended: 9,960 pairs/s where the forward
        alone does 15,600; round 4's extra staging buffers + device-to-device hop: 13,059 of 16,238).  The host buffer must stay untouched
        until its copy has run (rotate >= nplans + 2 pinned buffers, or wait for the events in `pipe.copied[n % pipe.nplans]`, one per copy stream)."""
        assert self.u8, "DetectionPipeline(u8=True) takes uint8 batches"
        pi = self.n % self.nplans
        fs = self.fwd_streams[pi % self.depth]
        dst = self.plans[pi].inputs[0]
        B, ncs = img6.shape[0], len(self.copy_streams)
        for k, cs in enumerate(self.copy_streams):                # the batch in `ncs` slices of images, one copy stream (DMA queue) each
            lo, hi = B * k // ncs, B * (k + 1) // ncs
            if hi <= lo:
                continue
            if self.n >= self.nplans:
                cs.wait_event(self.fwd_done[pi])                  # this plan's previous forward (nplans steps ago) has consumed its input
            if img6.is_cuda:
                cs.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(cs):
                dst[lo:hi].copy_(img6[lo:hi], non_blocking=True)
            if img6.is_cuda:
                img6.record_stream(cs)
            self.copied[pi][k].record(cs)
            fs.wait_event(self.copied[pi][k])
        return self.step()

    def step(self):
        """Enqueue one batch: forward replay, then NMS (+ gather) on the second stream.  Returns (det, count) device tensors, valid once
        the nms stream has drained (see synchronize()) — ALWAYS rank-major: det (world, B, max_det, 6) fp32, count (world, B) int32, i.e.
        global batch order for contiguous shards (dist.flatten_gathered gives the (world * B, ...) form).  Without a gather (one rank)
        the leading axis has length 1 and the tensors are views of the slot's NMS output block."""
        if self.nplans > 1:
            return self._step_deep()
        i = (self.n & 1) if self.overlap else 0
        fs, ns = self.fwd_stream, self.nms_stream
        self.plan.run(fs.cuda_stream)
        if self.overlap:
            if self.n >= 2:
                fs.wait_event(self.nms_done[i])             # NMS of step n-2 has finished reading zbuf[i]
            with torch.cuda.stream(fs):
                self.zbuf[i].copy_(self.z, non_blocking=True)
            self.snap_done[i].record(fs)
            ns.wait_event(self.snap_done[i])
        if self.gather and self.gathers:
            ns.wait_event(self.gather_done)                 # the last gather has read this slot's block
        det, count, keep = nms_device(self.zbuf[i], stream_ptr=ns.cuda_stream, runner=self.runners[i], **self.nms_args)
        out = (det[None], count[None])
        if self.gather:
            out = self._gather(self.runners[i].block, 0)
        if self.overlap:
            self.nms_done[i].record(ns)
        self.n += 1
        self.last = out
        return out

    def _gather(self, send, slot):
        """Step bookkeeping of the collective: returns the (det, count) views of this step inside the generation the NEXT gather writes, and
        issues that gather when the group is complete (or from synchronize(), for a group cut short).  `send` = what this rank contributes:
        one slot's block (one-plan pipeline) or the whole group_block."""
        B, max_det = self.runners[0].B, self.runners[0].max_det
        out = D.split_group_block(self.gathered[self.gen], self.world if self.world > 1 else 1, self.group, slot, B, max_det)
        self._send = send
        self.pending += 1
        if self.pending >= self.group:
            self._issue_gather()
        return out

    def _issue_gather(self):
        import torch.distributed as tdist
        ns, gs = self.nms_stream, self.gather_stream
        if gs is not ns:
            self.nms_group_done.record(ns)
            gs.wait_event(self.nms_group_done)              # behind the group's last NMS; the NMS stream itself goes on with the next batch
        with torch.cuda.stream(gs):
            tdist.all_gather_into_tensor(self.gathered[self.gen], self._send)
        self.gather_done.record(gs)
        self.gen ^= 1
        self.pending = 0
        self.gathers += 1

    def _step_deep(self):
        """Several plans: batch n runs plan n % nplans on forward stream n % depth (nplans is a multiple of depth: a plan keeps its stream); its NMS reads that plan's z directly (no snapshot: the
        plan is not replayed before its NMS has finished — which also orders the replay behind the plan's previous forward)."""
        pi = self.n % self.nplans
        fs, ns, plan = self.fwd_streams[pi % self.depth], self.nms_stream, self.plans[pi]
        if self.n >= self.nplans:
            fs.wait_event(self.nms_done_deep[pi])          # NMS of batch n - nplans has finished reading this plan's z
        plan.run(fs.cuda_stream)
        self.fwd_done[pi].record(fs)
        ns.wait_event(self.fwd_done[pi])
        if self.gather and self.gathers and self.pending == 0:
            ns.wait_event(self.gather_done)                # the last gather has read the blocks this group's NMS launches overwrite
        det, count, keep = nms_device(plan.outputs[0], stream_ptr=ns.cuda_stream, runner=self.deep_runners[pi], **self.nms_args)
        out = (det[None], count[None])
        if self.gather:
            out = self._gather(self.group_block, pi)
        self.nms_done_deep[pi].record(ns)
        self.n += 1
        self.last = out
        return out

    def synchronize(self):
        """Everything enqueued so far has finished.  With a process group this also sends a group that is not full yet (every rank has taken
        the same number of steps, so every rank issues the same collectives)."""
        if self.gather and self.pending:
            self._issue_gather()
        for fs in self.fwd_streams:
            fs.synchronize()
        self.nms_stream.synchronize()
        if self.gather_stream is not self.nms_stream:
            self.gather_stream.synchronize()

    def __call__(self, rgb, ir):
        """Convenience: copy one batch in, run it, wait, return list of (n, 6) detections per image (all ranks' images, global order)."""
        det, count = self.submit(rgb, ir)
        self.synchronize()
        det, count = D.flatten_gathered(det, count)
        return [det[k, :n].clone() for k, n in enumerate(count.tolist())]
