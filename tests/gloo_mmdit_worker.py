"""CPU worker (gloo): the orchestration of magcache_amd.mmdit.MMDiTSequenceParallel -- gather layout of the image K|V
shards, phase order, assembly of the sharded output -- with a recording host stand-in for the engine (no compute)."""
import json
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from magcache_amd import mmdit as MM  # noqa: E402
from magcache_amd._lib import MC_FAMILY_FLUX, MC_FAMILY_HUNYUAN, MC_MODE_FULL, MC_MODE_SKIP  # noqa: E402


class HostEngine:
    def __init__(self, family, rank, world):
        self.family, self.sp_rank, self.sp_size = family, rank, world
        self.img_tokens, self.tokens_per_rank, self.n_blocks = 8 * world, 8, 3
        self.out_channels, self.latent_grid, self.device = 4, (1, 4, 2 * world), torch.device("cpu")
        self.bufs = {"kv_gather": torch.zeros(world * 16 * 6, dtype=torch.bfloat16), "head_tokens": torch.zeros(16 * 64)}
        self.log, self.ok = [], True

    def buffer(self, name, dtype=None):
        return self.bufs[name]

    def begin(self, img, t, g, txt, txt_valid, vec, mode):
        self.log.append(("begin", int(mode)))

    def block_pre(self, blk):
        self.log.append(("pre", blk))
        kv = self.bufs["kv_gather"].view(self.sp_size, -1)
        kv.zero_()
        kv[self.sp_rank] = float(10 * blk + self.sp_rank + 1)          # this rank's shard of this block

    def block_attn_local(self, blk):
        self.log.append(("local", blk))

    def block_post(self, blk):
        self.log.append(("post", blk))
        kv = self.bufs["kv_gather"].view(self.sp_size, -1).float()
        want = torch.tensor([10.0 * blk + r + 1 for r in range(self.sp_size)])
        self.ok = self.ok and bool((kv == want[:, None]).all())           # every shard arrived in its slot

    def end(self, out):
        self.log.append(("end",))
        rows = torch.arange(self.tokens_per_rank).float()[:, None] + 100.0 * self.sp_rank
        if self.family == MC_FAMILY_HUNYUAN:
            self.bufs["head_tokens"].view(-1, 64)[:self.tokens_per_rank] = rows
        else:
            out.copy_(rows.expand(-1, self.out_channels))

    def unpatchify(self, tokens, out):
        self.log.append(("unpatchify",))
        out.copy_(tokens[:, 0].reshape(out.shape[1:]).expand_as(out))


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    res = {}
    for fam, name in ((MC_FAMILY_FLUX, "flux"), (MC_FAMILY_HUNYUAN, "hunyuan")):
        e = HostEngine(fam, rank, world)
        sp = MM.MMDiTSequenceParallel(e)
        out = sp.forward(None, 0.0, 0.0, None, 1, None, MC_MODE_FULL)
        want = torch.cat([torch.arange(8).float() + 100.0 * r for r in range(world)])
        got = out[:, 0] if fam == MC_FAMILY_FLUX else out[0].reshape(-1)
        full_log = list(e.log)
        e.log.clear()
        sp.forward(None, 0.0, 0.0, None, 1, None, MC_MODE_SKIP)
        res[name] = dict(out_ok=bool(torch.equal(got, want)), kv_ok=e.ok, log=full_log, skip_log=list(e.log))
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        json.dump(gathered, open(sys.argv[1], "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
