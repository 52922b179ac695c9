"""alias: main.py -> cosa_amd.main (same flags; `torchrun --nproc_per_node=N dropin/main.py EXP --dataset VOC12 ...`)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cosa_amd import args as cosa_args  # noqa: E402
from cosa_amd.main import finaleval, main  # noqa: E402,F401

if __name__ == "__main__":
    parsed, changed = cosa_args.parse()
    print("Changed arguments:")
    print(changed)
    main(parsed)
