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
CHECK = len(sys.argv) > 2 and sys.argv[2] == "check"       # also compare every launch shape with fp32 torch (chunks of 8192 rows)
ONLY = sys.argv[3] if len(sys.argv) > 3 else ""            # "teacher" | "student" | ""
TEACHER, STUDENT = 87904, 12560
SHAPES = [
    (TEACHER, 2304, 768, 0), (TEACHER, 768, 768, 2), (TEACHER, 3072, 768, 1), (TEACHER, 768, 3072, 2),
    (STUDENT, 2304, 768, 0), (STUDENT, 768, 768, 0), (STUDENT, 3072, 768, 0), (STUDENT, 768, 3072, 0), (STUDENT, 768, 2304, 0),
]


def main():
    dev = 'cuda'
    for M, N, K, epi in SHAPES:
        if (ONLY == "teacher" and M != TEACHER) or (ONLY == "student" and M != STUDENT):
            continue
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
        err = ""
        if CHECK:
            y = nn_ops.gemm_bf16(x, w, b, epi, res)
            worst = 0.0
            for r0 in range(0, M, 8192):
                ref = x[r0:r0 + 8192].float() @ w.float().t() + b.float()
                if epi == 1:
                    ref = torch.nn.functional.gelu(ref)
                if epi == 2:
                    ref = ref + res[r0:r0 + 8192]
                worst = max(worst, ((y[r0:r0 + 8192].float() - ref).abs().max() / ref.abs().max().clamp_min(1.0)).item())
            err = "  max err / scale %.2e" % worst
        print("M=%6d N=%5d K=%5d epi=%d  %8.1f us  %7.1f TFLOP/s%s" % (M, N, K, epi, us, 2.0 * M * N * K / us * 1e-6, err), flush=True)


if __name__ == '__main__':
    main()
