"""Steps in flight: `depth` hipGraph instances of the forward on `depth` streams.

One forward of this path leaves most of the chip idle -- farthest point sampling holds ONE CU per cloud for two thirds
of the step -- so a server that has the next batch ready overlaps consecutive batches: batch i runs on
slot i % depth while the previous depth-1 batches are still in their latency-bound phases.  The reference has no
counterpart (a TF session runs one `sess.run` at a time, core/model.py:135-210 builds one graph); every slot computes
exactly that graph: a slot's index outputs (kNN, FPS, three_nn) are bit-equal to the serial forward of the same batch
and its descriptors agree within 2e-6 (with steps in flight the model places three_nn differently and runs the local
tail as one launch -- the same products in another association; tests/test_engine_gpu.py).

    pipe = model.pipeline(example_points, depth=4, outputs=("xyz_feat",))
    t = pipe.submit(batch)            # copies `batch` into the slot's input buffer and replays the slot's graph
                                      # (a pinned HOST batch: async H2D on the slot's own stream; fetch_to=: D2H behind it)
    ...                               # up to depth-1 further submits before the slot is reused
    outs = pipe.result(t)             # the CURRENT stream waits for that step; dict of the slot's static buffers

Zero-copy hand-over: write the batch into `pipe.input_buffer(slot)` ON `pipe.stream(slot)` (e.g. as the destination of
the host-to-device copy enqueued there) and call `pipe.submit()` without an argument; a batch written on the current
stream instead needs `pipe.submit(after_current=True)`.
"""
import torch


class Ticket:
    """One submitted step: the slot it ran on and the event that marks its end."""
    __slots__ = ("slot", "event", "seq")

    def __init__(self, slot, event, seq):
        self.slot, self.event, self.seq = slot, event, seq


class Pipeline:
    """`depth` captured forwards of ONE model, each with its own batch buffers and stream.

    The persistent flex_conv kernels are told how many other steps' FPS kernels hold CUs while they run
    (`steps_in_flight` -> `Geometry.busy_cus_per_xcd` -> `reserve_cus_per_xcd`, a placement hint: speed only, the
    results do not depend on it -- tests/test_engine_gpu.py::test_flex_conv_x6_reserve_hint_vs_oracle)."""

    def __init__(self, model, example_points, depth=2, outputs=None, example_knn=None, streams=None, warmup=2):
        if depth < 1:
            raise ValueError("depth must be >= 1")
        if streams is not None and len(streams) < depth:
            raise ValueError("need %d streams, got %d" % (depth, len(streams)))
        self.model, self.depth, self.outputs = model, int(depth), outputs
        dev = example_points.device
        self._streams = list(streams[:depth]) if streams is not None else [torch.cuda.Stream(device=dev) for _ in range(depth)]
        hint = getattr(model, "steps_in_flight", 1)
        model.steps_in_flight = self.depth  # read by DH3D._geometry while the slots are captured
        try:
            with torch.no_grad():
                self._runs = [model.graphed(example_points, example_knn, outputs=outputs, warmup=warmup if k == 0 else 1)
                              for k in range(self.depth)]
        finally:
            model.steps_in_flight = hint
        self._version = model.weights_version
        # one event per step of the last TWO rounds over the slots: a ticket's event is not re-recorded until 2 * depth
        # further submits, so a host-side consumer may lag a whole round behind the submitting loop without blocking it
        self._events = [torch.cuda.Event() for _ in range(2 * self.depth)]
        self._consumed = [None] * self.depth  # event after which slot k's buffers may be overwritten
        self._seq = 0
        cur = torch.cuda.current_stream(dev)
        for st in self._streams:
            st.wait_stream(cur)

    # ------------------------------------------------------------------ slots
    def stream(self, slot):
        return self._streams[slot]

    def input_buffer(self, slot):
        """The slot's resident batch [Bt, N, 3] for a zero-copy step.  WHERE the write is enqueued matters: submit() without
        `points` does NOT order the slot's stream behind the caller's current stream, so write the next batch here ON
        `stream(slot)` (`with torch.cuda.stream(pipe.stream(k)): buf.copy_(host_batch, non_blocking=True)` -- the copy then
        also waits for the slot's previous step by stream order), or write it on the current stream and call
        `submit(after_current=True)`.  A write on the current stream followed by a plain `submit()` races the graph replay
        against it."""
        return self._runs[slot].static_input

    def knn_buffer(self, slot):
        return self._runs[slot].static_knn

    @property
    def next_slot(self):
        return self._seq % self.depth

    # ------------------------------------------------------------------ steps
    def submit(self, points=None, knn_inds=None, after_current=None, fetch_to=None):
        """Enqueue one full forward on the next slot.  `points` (optional) is copied into the slot's input buffer on the
        slot's stream, after the caller's stream has reached this point (so a batch produced on the current stream is
        complete) and after the previous consumer of this slot's outputs (result()) is done with them.

        HOST batches (the serving loop: localdesc_extract.py:106-138, globaldesc_extract.py:84-100 feed numpy arrays): a
        pinned CPU tensor is read by a staging kernel (dh3d_stage_copy) ON THE SLOT'S STREAM in front of the replay --
        nothing on the device produced it, so the slot's stream is NOT ordered behind the current one (no event record +
        wait per submit), and no copy engine sits on the step's chain (hipMemcpyAsync H2D on the slot's stream: 0.246 ->
        0.407 ms per step of the local pipeline, tools/streaming_probe.py).  The caller keeps the pinned buffer
        untouched until the step's ticket has completed.
        `fetch_to` ({output name: pinned CPU tensor or device tensor}) appends the copies of those outputs on the same
        stream (same kernel), in front of the ticket's event: `ticket.event.synchronize()` then means "the descriptors
        are in host memory".

        Zero-copy (no `points`): the batch must already be in `input_buffer(slot)`, written ON `stream(slot)` -- or on
        the current stream with `after_current=True`, which orders the slot's stream behind the current one first (an
        event record + wait per submit: measured 31.1 k -> 26.4 k clouds/s on the local workload four deep, which is why
        it is not the default for the zero-copy path).  With DEVICE `points` given the order is always established."""
        if self.model.weights_version != self._version:
            raise RuntimeError("the model's weights changed (optimiser step / invalidate / load_state_dict) after this "
                               "pipeline was captured: build a new one")
        k = self._seq % self.depth
        run, st = self._runs[k], self._streams[k]
        from_device = any(t is not None and t.is_cuda for t in (points, knn_inds))
        if after_current or (after_current is None and from_device):
            st.wait_stream(torch.cuda.current_stream(run.static_input.device))
        # the copies below run on the slot's stream: tell the caching allocator, or a caller that drops `points`
        # right after submit() may see its block handed out again (on ITS stream) before the copy has read it
        for t in (points, knn_inds):
            if t is not None and t.is_cuda:
                t.record_stream(st)
        if self._consumed[k] is not None:
            st.wait_event(self._consumed[k])
            self._consumed[k] = None
        with torch.cuda.stream(st):
            outs = run(points, knn_inds)
            if fetch_to:
                from . import pm
                for name, dst in fetch_to.items():
                    src = outs[name]
                    if ((dst.is_cuda or dst.is_pinned()) and dst.is_contiguous() and src.is_contiguous()
                            and dst.dtype == src.dtype and dst.numel() == src.numel()):
                        # a kernel on this stream (pinned host memory is written over the link).  Also for a device
                        # destination: behind a host batch's staging kernel a tensor copy_ of the 34 MB local map on the
                        # same stream took the step from 0.25 to 4.6 ms (bench.py value_streaming, round 6)
                        pm.stage_copy(src, dst)
                    else:
                        dst.copy_(src, non_blocking=True)
            ev = self._events[self._seq % (2 * self.depth)]
            ev.record(st)
        t = Ticket(k, ev, self._seq)
        self._seq += 1
        return t

    def result(self, ticket, wait="stream"):
        """The outputs of a submitted step (the slot's static buffers: valid until `depth` further submits).
        wait="stream": the current stream waits for the step (no host block); "host": the host does; None: neither."""
        if self._seq - ticket.seq > self.depth:
            raise RuntimeError("slot %d has been reused since this step was submitted" % ticket.slot)
        if wait == "stream":
            torch.cuda.current_stream().wait_event(ticket.event)
        elif wait == "host":
            ticket.event.synchronize()
        return self._runs[ticket.slot].outputs

    def release(self, ticket):
        """Mark the current stream's position as the end of the consumer's reads of this step's outputs (call after
        the kernels that read them were enqueued): the slot's next step waits for it."""
        ev = torch.cuda.Event()
        ev.record()
        self._consumed[ticket.slot] = ev

    def drain(self):
        """The current stream waits for every slot (the host does not)."""
        cur = torch.cuda.current_stream()
        for st in self._streams:
            cur.wait_stream(st)

    def map(self, batches, clone=True):
        """Run an iterable of batches (device tensors or pinned host tensors) `depth` deep and yield their output dicts in
        order.  clone=True (default): every step's outputs are copied out of the slot's buffers ON THE SLOT'S STREAM right
        behind the replay (a slot's buffers are overwritten `depth` steps later), and the current stream waits for that
        step's event before the dict is yielded -- one cross-stream dependency per step.  clone=False yields the slot's
        static buffers themselves (valid until `depth` further submits) and orders the slot's next step behind the
        current stream's position at the time the NEXT item is requested (release)."""
        tickets = []
        for b in batches:
            if len(tickets) == self.depth:
                yield from self._take(*tickets.pop(0), clone)
            tickets.append(self._submit_for_map(b, clone))
        while tickets:
            yield from self._take(*tickets.pop(0), clone)

    def _submit_for_map(self, b, clone):
        if not clone:
            return self.submit(b), None
        k = self.next_slot
        with torch.cuda.stream(self._streams[k]):   # (allocated on the slot's stream: the copy below is their first use)
            copies = {name: torch.empty_like(v) for name, v in self._runs[k].outputs.items()}
        return self.submit(b, fetch_to=copies), copies

    def _take(self, ticket, copies, clone):
        cur = torch.cuda.current_stream()
        if not clone:
            yield self.result(ticket, wait="stream")
            self.release(ticket)   # (runs when the consumer asks for the next item: its reads are enqueued by then)
            return
        cur.wait_event(ticket.event)
        for v in copies.values():
            v.record_stream(cur)   # handed from the slot's stream to the consumer's
        yield copies
