"""What HBM bandwidth a plain streaming kernel reaches on this box (the practical roof beside the 8 TB/s of the data sheet):
device-to-device copy (read + write), fill (write only), sum (read only), 4 GiB buffers, HIP-event timing of 10 repeats.
Usage (GPU box): python3 scripts/hbm_probe.py"""
import torch


def timed(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = 1 << 30                                  # 4 GiB of float32
    x = torch.empty(n, dtype=torch.float32, device='cuda').fill_(1.0)
    y = torch.empty_like(x)
    gb = n * 4 / 1e9
    t = timed(lambda: y.copy_(x))
    print('copy  (read + write): %.3f ms  %.0f GB/s' % (t * 1e3, 2 * gb / t))
    t = timed(lambda: y.fill_(2.0))
    print('fill  (write only):   %.3f ms  %.0f GB/s' % (t * 1e3, gb / t))
    t = timed(lambda: x.sum())
    print('sum   (read only):    %.3f ms  %.0f GB/s' % (t * 1e3, gb / t))
    h = x.view(torch.int32)
    t = timed(lambda: torch.bitwise_or(h, 1, out=y.view(torch.int32)))
    print('or    (read + write): %.3f ms  %.0f GB/s' % (t * 1e3, 2 * gb / t))


if __name__ == '__main__':
    main()
