"""alias: the part of utils/misc.py the launcher touches (init_distributed_mode :405-445) -> cosa_amd.main"""
from cosa_amd.main import init_distributed_mode  # noqa: F401


def get_sha():
    return "cosa_amd"


def is_main_process():
    import torch.distributed as dist
    return not (dist.is_available() and dist.is_initialized()) or dist.get_rank() == 0
