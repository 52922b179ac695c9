"""Time the projection GEMM on the launch shapes of one training step (teacher: all tokens of the 3 scales x 2 flips; student: 16 x 785).

    python tools/gemm_shapes.py [reps]

Prints one line per (M, N, K, epilogue): average microseconds over `reps` back-to-back launches (events on the current stream) and
TFLOP/s.  Runs unchanged in an older checkout of this repository (same nn_ops.gemm_bf16 signature), which is how two rounds are
compared on one box.
"""
import sys
import torch

sys.path.insert(0, '.')
from cosa_amd import nn_ops  # noqa: E402

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
TEACHER, STUDENT = 87904, 12560
SHAPES = [
    (TEACHER, 2304, 768, 0), (TEACHER, 768, 768, 2), (TEACHER, 3072, 768, 1), (TEACHER, 768, 3072, 2),
    (STUDENT, 2304, 768, 0), (STUDENT, 768, 768, 0), (STUDENT, 3072, 768, 0), (STUDENT, 768, 3072, 0), (STUDENT, 768, 2304, 0),
]


def main():
    dev = 'cuda'
    for M, N, K, epi in SHAPES:
        x = torch.randn(M, K, device=dev).bfloat16()
        w = (torch.randn(N, K, device=dev) * 0.03).bfloat16()
        b = torch.randn(N, device=dev).bfloat16()
        res = torch.randn(M, N, device=dev) if epi == 2 else None
        for _ in range(3):
            nn_ops.gemm_bf16(x, w, b, epi, res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REPS):
            nn_ops.gemm_bf16(x, w, b, epi, res)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / REPS
        print("M=%6d N=%5d K=%5d epi=%d  %8.1f us  %7.1f TFLOP/s" % (M, N, K, epi, us, 2.0 * M * N * K / us * 1e-6), flush=True)


if __name__ == '__main__':
    main()
