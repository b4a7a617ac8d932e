"""Does Event.synchronize() return when the EVENT is reached, or when the stream it was recorded on has drained?
kernel A (~5 ms) -> D2H copy -> event -> kernel B (~5 ms), all on one stream; the host waits for the event.  Second form: the copy and
its event on a stream of their own (which waits for A through an event)."""
import time
import torch

dev = "cuda"
x = torch.randn(8192, 8192, device=dev)
small = torch.zeros(4, 513, 513, dtype=torch.uint8, device=dev)
host = torch.empty(small.shape, dtype=torch.uint8, pin_memory=True)


def busy(n):
    y = x
    for _ in range(n):
        y = y @ x
    return y


busy(2); torch.cuda.synchronize()
t0 = time.perf_counter(); busy(8); torch.cuda.synchronize(); per = (time.perf_counter() - t0) / 8
n = max(1, int(5e-3 / per))
print("one matmul %.3f ms; A = B = %d matmuls" % (per * 1e3, n))
for form in ("same stream", "copy stream"):
    for rep in range(3):
        torch.cuda.synchronize()
        main = torch.cuda.current_stream()
        t0 = time.perf_counter()
        busy(n)
        if form == "same stream":
            host.copy_(small, non_blocking=True)
            ev = torch.cuda.Event(); ev.record()
        else:
            cs = getattr(busy, "cs", None) or torch.cuda.Stream()
            busy.cs = cs
            cs.wait_stream(main)
            with torch.cuda.stream(cs):
                host.copy_(small, non_blocking=True)
                ev = torch.cuda.Event(); ev.record()
        busy(n)
        t1 = time.perf_counter()
        ev.synchronize()
        t2 = time.perf_counter()
        torch.cuda.synchronize()
        t3 = time.perf_counter()
        print("%-12s enqueue %.2f ms, event reached at %.2f ms, stream drained at %.2f ms" % (form, 1e3 * (t1 - t0), 1e3 * (t2 - t0), 1e3 * (t3 - t0)))
