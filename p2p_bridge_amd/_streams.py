"""A side stream that really runs BESIDE another one.

HIP maps every stream to one of GPU_MAX_HW_QUEUES hardware queues (4 by default) in creation order, and two streams on one queue
execute one after the other, whatever the program says. Which queue a `torch.cuda.Stream()` lands on depends on how many streams
the process created before -- measured in round 6: the training loop's alignment prefetch (train.AlignedBatches) overlapped the
captured step with 2 and 16 queues and ran strictly behind it with 4 and 8 (profiles/r06_train_align_ab.txt), purely by the
position of its stream in torch's pool. `concurrent_stream()` therefore PROBES: a few hundred microseconds of work on the
reference stream, one small kernel on the candidate, device timestamps; the first candidate whose kernel finishes while the
reference stream is still busy is returned. (The sampler's two chain streams were probed the same way and gain nothing: a
replayed hipGraph is scheduled on the runtime's own queues -- 2 / 4 / 8 hardware queues x probed / unprobed chain streams: equal,
profiles/r06_train_align_ab.txt (4).)"""
import torch

_probe_buf = {}


def runs_beside(ref: torch.cuda.Stream, cand: torch.cuda.Stream) -> bool:
    dev = ref.device
    if dev not in _probe_buf:
        _probe_buf[dev] = (torch.zeros(32 << 20, device=dev), torch.zeros(1024, device=dev))
    big, small = _probe_buf[dev]
    torch.cuda.synchronize(dev)
    t0, t_ref, t_cand = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(ref):
        t0.record(ref)
        for _ in range(24):  # ~0.5-1 ms of memory-bound work on the reference stream
            big.add_(1.0)
        t_ref.record(ref)
    cand.wait_event(t0)
    with torch.cuda.stream(cand):
        small.add_(1.0)
        t_cand.record(cand)
    torch.cuda.synchronize(dev)
    return t0.elapsed_time(t_cand) < 0.5 * t0.elapsed_time(t_ref)


def concurrent_stream(ref: torch.cuda.Stream = None, tries: int = 12) -> torch.cuda.Stream:
    """a new stream whose work overlaps `ref`'s (default: the current stream); after `tries` candidates the last one is returned
    (correct either way -- only the overlap is lost)"""
    ref = ref if ref is not None else torch.cuda.current_stream()
    cand = None
    for _ in range(max(1, tries)):
        cand = torch.cuda.Stream(device=ref.device)
        if cand != ref and runs_beside(ref, cand) and runs_beside(ref, cand):  # (twice: one lucky timing is not a queue)
            return cand
    return cand

