#!/usr/bin/env python3
"""Development micro-benchmark of call_tiles_kernel (kernel time from HIP events, no CPU baseline).
PISCES_HIP_LIB selects an experimental build. Prints one line per configuration."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loci", type=int, default=100_000)
    ap.add_argument("--depth", type=int, default=500)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--ring", type=int, default=4)
    ap.add_argument("--tag", default="")
    ap.add_argument("--tile", type=int, default=64, help="loci per tile of the synthetic tuple stream")
    ap.add_argument("--plan", default="", help="tiles of unequal size: '1536x58,256x43' = the first 1536 tiles of 58 loci, the next 256 of 43, the rest --tile")
    a = ap.parse_args()
    plan = [(int(x.split("x")[0]), int(x.split("x")[1])) for x in a.plan.split(",") if x] or None
    import torch
    from pisces_amd import _abi, engine, synth
    dev = torch.device("cuda", 0)
    ring = [synth.make_pileup(a.loci, a.depth, seed=100 + b, device=dev, tile=a.tile, tile_plan=plan) for b in range(a.ring)]
    for p in ring:
        p.base = p.qual = None
    torch.cuda.empty_cache()
    nt = ring[0].n_tiles
    cap = nt * 256
    rec = torch.zeros(cap * 64, dtype=torch.uint8, device=dev)
    tr = torch.zeros(nt * 48, dtype=torch.uint8, device=dev)
    st = None   # the handle's own stream (fills are synchronised before the first launch)
    with engine.HipVariantCaller(_abi.default_config()) as c:
        def step(i):
            p = ring[i % a.ring]
            c.call_tiles(p.tuples.data_ptr(), p.tiles.data_ptr(), p.n_tiles, p.ref.data_ptr(), 1, p.ref_len,
                         rec.data_ptr(), cap, tr.data_ptr(), st)
        torch.cuda.synchronize()
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        c.set_timing(True)
        for i in range(a.steps):
            step(i)
        torch.cuda.synchronize()
        ms, n = c.kernel_time()
    k = ms / n
    nb = 4 * ring[0].n_obs + a.loci + 64 * a.loci
    print(f"{a.tag or os.environ.get('PISCES_HIP_LIB', 'product')}: loci={a.loci} depth={a.depth} tile={a.tile} plan={a.plan or '-'} tiles={nt} kernel={k*1e3:.1f} us  "
          f"{nb / (k * 1e-3) / 1e9:.0f} GB/s algorithmic ({nb / (k * 1e-3) / 8e12 * 100:.1f}% of 8 TB/s)  "
          f"{a.loci / (k * 1e-3) / 1e9:.2f} G loci/s", flush=True)


if __name__ == "__main__":
    main()
