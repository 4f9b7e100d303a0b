"""Ranks started by ``torch.distributed.run`` (or any launcher that sets RANK /
LOCAL_RANK / WORLD_SIZE) for ``woltka_amd.workflow.workflow(comm=...)``:
profiles gathered with gloo on every rank.  An adapter outside the package --
``woltka_amd`` itself starts its ranks with multiprocessing
(``shard.LocalWorld``) and never imports PyTorch.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \\
        tools/torch_world.py -- <arguments of `woltka classify`>
"""
import os
import sys


def env_rank():
    """(rank, local_rank, world) from the launcher's environment."""
    return (int(os.environ.get('RANK', '0')),
            int(os.environ.get('LOCAL_RANK', '0')),
            int(os.environ.get('WORLD_SIZE', '1')))


class TorchWorld:
    kind = 'torch'

    def __init__(self):
        self.rank, self.local, self.world = env_rank()

    def gather(self, obj):
        import torch.distributed as dist
        if not dist.is_initialized():
            os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
            dist.init_process_group('gloo')
        out = [None] * dist.get_world_size()
        dist.all_gather_object(out, obj)
        return out


def main(argv):
    sys.path.insert(0, os.path.dirname(os.path.dirname(
        os.path.abspath(__file__))))
    from woltka_amd.cli import classify_cmd
    ctx = classify_cmd.make_context('classify', [a for a in argv if a != '--'])
    from woltka_amd.workflow import workflow
    workflow(comm=TorchWorld(), **ctx.params)


if __name__ == '__main__':
    main(sys.argv[1:])
