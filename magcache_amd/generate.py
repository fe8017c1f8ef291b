"""`python -m magcache_amd.generate` -- the reference's `magcache_generate.py` entry point on the HIP engine.

Same flags, defaults and argument validation as MagCache4Wan2.1/magcache_generate.py (`_parse_args`
:598-775, `_validate_args` :563-595) for everything that reaches the denoising hot path:

    --task --size --frame_num --ckpt_dir --base_seed --sample_solver --sample_steps --sample_shift
    --sample_guide_scale --use_magcache --magcache_thresh --magcache_K --retention_ratio
    --magcache_calibration --save_file --prompt

What is outside the hot path and absent offline is declared, not faked: there is no T5 text encoder and
no VAE in this repository (out of scope).  So
  * the text context is read from `--context_file` / `--context_null_file` (torch tensors [len<=512, 4096],
    e.g. dumped from the upstream pipeline), or, without them, is a seeded synthetic stand-in;
  * DiT weights are loaded from `--ckpt_dir` if it holds the upstream `*.safetensors`, otherwise they are
    seeded random-init weights of the named architecture (a warning says so);
  * i2v-14B takes `--clip_fea_file` ([257, 1280] CLIP features of the first frame) and `--y_file` ([20, F, H/8, W/8]:
    mask + VAE latent of the conditioning frames), vace-* takes `--vace_context_file` ([96, F, H/8, W/8]) and
    `--vace_context_scale`: the tensors the upstream pipeline computes with its CLIP / VAE before the sampling loop
    (synthetic stand-ins without the files);
  * the result saved to `--save_file` is the final LATENT ([16, F, H/8, W/8] fp32, torch.save), which the
    upstream pipeline would hand to its VAE decoder.
The flags that only concern those parts (--t5_cpu, --offload_model, prompt extension, FSDP, ...) are accepted
and ignored so that existing command lines keep working.  Multi-GPU: launch with torchrun; the two CFG
branches run on two halves of the node and the token sequence is sharded inside each half
(magcache_amd/parallel.py) instead of the reference's xfuser USP.
"""
import argparse
import glob
import hashlib
import logging
import os
import random
import sys
import time

# upstream wan/configs: SIZE_CONFIGS / SUPPORTED_SIZES for the tasks this engine implements
SIZE_CONFIGS = {"720*1280": (720, 1280), "1280*720": (1280, 720), "480*832": (480, 832), "832*480": (832, 480),
                "1024*1024": (1024, 1024), "704*1280": (704, 1280), "1280*704": (1280, 704)}
SUPPORTED_SIZES = {"t2v-14B": ("720*1280", "1280*720", "480*832", "832*480"), "t2v-1.3B": ("480*832", "832*480"),
                   "t2i-14B": tuple(SIZE_CONFIGS.keys()),
                   "i2v-14B": ("720*1280", "1280*720", "480*832", "832*480"),
                   "vace-1.3B": ("480*832", "832*480"), "vace-14B": ("720*1280", "1280*720", "480*832", "832*480"),
                   # Wan2.2 (MagCache4Wan2.2/magcache_generate.py): the two-expert A14B models and the dense TI2V-5B
                   "t2v-A14B": ("720*1280", "1280*720", "480*832", "832*480"),
                   "i2v-A14B": ("720*1280", "1280*720", "480*832", "832*480"),
                   "ti2v-5B": ("704*1280", "1280*704")}
EXAMPLE_PROMPT = "Two anthropomorphic cats in comfy boxing gear and bright gloves fight intensely on a spotlighted stage."


def str2bool(v):
    if isinstance(v, bool):
        return v
    if v.lower() in ("yes", "true", "t", "y", "1"):
        return True
    if v.lower() in ("no", "false", "f", "n", "0"):
        return False
    raise argparse.ArgumentTypeError("Boolean value expected (True/False)")


def _validate_args(args):
    """magcache_generate.py:563-595"""
    assert args.task in SUPPORTED_SIZES, f"Unsupport task: {args.task} (this engine: {', '.join(SUPPORTED_SIZES)})"
    if args.task.endswith("A14B") or args.task == "ti2v-5B":
        # MagCache4Wan2.2/magcache_generate.py:409-419: the defaults come from the task's upstream config
        from magcache_amd.wan22 import WAN22_DEFAULTS
        d = WAN22_DEFAULTS[args.task]
        args.sample_steps = d["sample_steps"] if args.sample_steps is None else args.sample_steps
        args.sample_shift = d["sample_shift"] if args.sample_shift is None else args.sample_shift
        g = args.sample_guide_scale
        if args.task == "ti2v-5B":
            args.sample_guide_scale = d["guide_scale"] if g is None else g
            args.frame_num = d["frame_num"] if args.frame_num is None else args.frame_num
        else:
            args.sample_guide_scale = d["guide_scale"] if g is None else (g, g)     # (low-noise, high-noise expert)
    if args.sample_guide_scale is None:
        args.sample_guide_scale = 5.0
    if args.sample_steps is None:
        args.sample_steps = 50
    if args.sample_shift is None:
        args.sample_shift = 5.0
    if args.frame_num is None:
        args.frame_num = 1 if "t2i" in args.task else 81
    if "t2i" in args.task:
        assert args.frame_num == 1, f"Unsupport frame_num {args.frame_num} for task {args.task}"
    args.base_seed = args.base_seed if args.base_seed >= 0 else random.randint(0, sys.maxsize)
    assert args.size in SUPPORTED_SIZES[args.task], \
        f"Unsupport size {args.size} for task {args.task}, supported sizes are: {', '.join(SUPPORTED_SIZES[args.task])}"
    assert (args.frame_num - 1) % 4 == 0, "frame_num must be 4n+1 (VAE temporal stride 4)"


def _parse_args(argv=None):
    p = argparse.ArgumentParser(description="Denoise a Wan2.1 latent with MagCache on the MI355X HIP engine")
    p.add_argument("--task", type=str, default="t2v-14B", choices=list(SUPPORTED_SIZES.keys()))
    p.add_argument("--size", type=str, default="1280*720", choices=list(SIZE_CONFIGS.keys()))
    p.add_argument("--frame_num", type=int, default=None)
    p.add_argument("--ckpt_dir", type=str, default=None)
    p.add_argument("--save_file", type=str, default=None)
    p.add_argument("--prompt", type=str, default=None)
    p.add_argument("--base_seed", type=int, default=-1)
    p.add_argument("--sample_solver", type=str, default="unipc", choices=["unipc", "dpm++", "euler"])
    p.add_argument("--sample_steps", type=int, default=None)
    p.add_argument("--sample_shift", type=float, default=None)
    p.add_argument("--sample_guide_scale", type=float, default=None)
    p.add_argument("--magcache_thresh", type=float, default=0.12)
    p.add_argument("--retention_ratio", type=float, default=0.2)
    p.add_argument("--magcache_K", type=int, default=2)
    p.add_argument("--use_magcache", action="store_true", default=False)
    p.add_argument("--magcache_calibration", action="store_true", default=False)
    # inputs that replace the absent text encoder
    p.add_argument("--context_file", type=str, default=None, help="torch tensor [len<=512, 4096]: T5 embedding of the prompt")
    p.add_argument("--context_null_file", type=str, default=None, help="the same for the negative prompt")
    p.add_argument("--clip_fea_file", type=str, default=None, help="i2v: torch tensor [257, 1280] (CLIP visual features)")
    p.add_argument("--y_file", type=str, default=None, help="i2v: torch tensor [20, F, H/8, W/8] (mask + conditioning latent)")
    p.add_argument("--vace_context_file", type=str, default=None, help="vace: torch tensor [96, F, H/8, W/8]")
    p.add_argument("--vace_context_scale", type=float, default=1.0)
    # accepted for command-line compatibility; they configure parts that are not in this repository
    for flag, kw in (("--offload_model", dict(type=str2bool, default=None)), ("--ulysses_size", dict(type=int, default=1)),
                     ("--ring_size", dict(type=int, default=1)), ("--t5_fsdp", dict(action="store_true")),
                     ("--t5_cpu", dict(action="store_true")), ("--dit_fsdp", dict(action="store_true")),
                     ("--use_prompt_extend", dict(action="store_true")),
                     ("--prompt_extend_method", dict(type=str, default="local_qwen")),
                     ("--prompt_extend_model", dict(type=str, default=None)),
                     ("--prompt_extend_target_lang", dict(type=str, default="zh")),
                     ("--image", dict(type=str, default=None)), ("--src_video", dict(type=str, default=None)),
                     ("--src_mask", dict(type=str, default=None)), ("--src_ref_images", dict(type=str, default=None)),
                     ("--first_frame", dict(type=str, default=None)), ("--last_frame", dict(type=str, default=None))):
        p.add_argument(flag, **kw)
    args = p.parse_args(argv)
    _validate_args(args)
    return args


def _context(path, prompt, seed, text_dim, device):
    import torch
    if path:
        c = torch.load(path, map_location="cpu")
        c = c[0] if isinstance(c, (list, tuple)) else c
        assert c.dim() == 2 and c.shape[0] <= 512 and c.shape[1] == text_dim, f"context {tuple(c.shape)}"
        return c.float().to(device)
    h = int(hashlib.sha256((prompt or "").encode()).hexdigest()[:8], 16)
    g = torch.Generator(device="cpu").manual_seed(seed ^ h)
    n = max(8, min(512, len((prompt or "").split()) * 2 + 2))   # a stand-in of plausible length, zero padded by the engine
    return torch.randn(n, text_dim, generator=g).to(device)


def generate(args):
    import torch
    import torch.distributed as dist
    import magcache_amd as mca
    from magcache_amd import model as M
    from magcache_amd.engine import (WAN_I2V_14B, WAN_T2V_1_3B, WAN_T2V_14B, WAN_VACE_1_3B, WAN_VACE_14B,
                                     synthetic_weights)

    rank, world = int(os.getenv("RANK", 0)), int(os.getenv("WORLD_SIZE", 1))
    local = int(os.getenv("LOCAL_RANK", 0))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.ERROR,
                        format="[%(asctime)s] %(levelname)s: %(message)s", handlers=[logging.StreamHandler(sys.stdout)])
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device(device))
        seed = [args.base_seed] if rank == 0 else [None]
        dist.broadcast_object_list(seed, src=0)      # magcache_generate.py:852-855
        args.base_seed = seed[0]
    layout = None
    if world > 1:
        from magcache_amd.parallel import ParallelLayout
        layout = ParallelLayout(cfg_parallel=not args.magcache_calibration)   # calibration needs both branches per rank
        logging.info(f"parallel layout: {layout.describe()}")
    if args.ulysses_size > 1 or args.ring_size > 1:
        logging.info("--ulysses_size/--ring_size are ignored: the token sequence is sharded over WORLD_SIZE ranks")

    if args.task.endswith("A14B"):
        return _generate_wan22(args, device, rank, world, layout)
    if args.task == "ti2v-5B":
        return _generate_ti2v(args, device, rank, world, layout)
    is_i2v, is_vace = "i2v" in args.task, "vace" in args.task
    if is_i2v:
        cfg = WAN_I2V_14B
    elif is_vace:
        cfg = WAN_VACE_1_3B if "1.3B" in args.task else WAN_VACE_14B
    else:
        cfg = WAN_T2V_1_3B if "1.3B" in args.task else WAN_T2V_14B
    H, W = SIZE_CONFIGS[args.size][1], SIZE_CONFIGS[args.size][0]
    grid = ((args.frame_num - 1) // 4 + 1, H // 8, W // 8)
    logging.info(f"Generation job args: {args}")
    logging.info(f"latent grid {grid}, {grid[0] * (grid[1] // 2) * (grid[2] // 2)} tokens, {cfg['num_layers']} layers d={cfg['dim']}")
    model = M.WanModelHIP(cfg, grid, device=device, calibration=args.magcache_calibration,
                          sp_rank=layout.sp_rank if layout else 0, sp_size=layout.sp_size if layout else 1,
                          sp_group=layout.sp_group if layout else None)

    files = sorted(glob.glob(os.path.join(args.ckpt_dir or "", "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        sd = {}
        for f in files:
            sd.update(load_file(f, device="cpu"))
        model.load_state_dict(sd)
        logging.info(f"loaded {len(sd)} tensors from {args.ckpt_dir}")
    else:
        logging.warning("no *.safetensors under --ckpt_dir: using seeded RANDOM-INIT weights of the architecture")
        model.engine.load_weights(synthetic_weights(cfg, seed=0, device=device))

    if args.use_magcache:
        name = args.ckpt_dir or ("Wan2.1-T2V-1.3B" if "1.3B" in args.task else "Wan2.1-T2V-14B")
        if "T2V-1.3B" not in name and "T2V-14B" not in name:
            name = "Wan2.1-T2V-1.3B" if "1.3B" in args.task else "Wan2.1-T2V-14B"
        hint = args.ckpt_dir or (("14B" if "14B" in args.task else "1.3B") if is_vace else
                                 ("720P" if "720" in args.size else "480P"))
        table = M.select_table(hint, task=args.task) if (is_i2v or is_vace) else None    # :1001-1004, :1141-1144
        mca.init_magcache(model, args.sample_steps, args.magcache_thresh, args.magcache_K, args.retention_ratio,
                          ckpt_dir=name, mag_ratios=table)                        # :896-919
        if is_vace:
            type(model).forward = M.magcache_vace_forward                         # :1127
    if args.magcache_calibration:
        mca.init_magcache_calibration(model, args.sample_steps)                   # :921-928
        if is_vace:
            type(model).forward = M.magcache_vace_calibration                     # :1153
    if is_vace and not (args.use_magcache or args.magcache_calibration):
        type(model).forward = M.vace_plain_forward

    prompt = args.prompt or EXAMPLE_PROMPT
    ctx = _context(args.context_file, prompt, args.base_seed, cfg["text_dim"], device)
    ctx_null = _context(args.context_null_file, "", args.base_seed + 1, cfg["text_dim"], device)
    g = torch.Generator(device=device).manual_seed(args.base_seed)
    noise = torch.randn(16, *grid, dtype=torch.float32, device=device, generator=g)      # wan_magcache.py:243-251
    extra = {}
    if is_i2v:
        gi = torch.Generator(device="cpu").manual_seed(args.base_seed + 2)
        extra["clip_fea"] = (torch.load(args.clip_fea_file, map_location="cpu") if args.clip_fea_file
                             else torch.randn(257, cfg["clip_dim"], generator=gi)).float().to(device)
        extra["y"] = [(torch.load(args.y_file, map_location="cpu") if args.y_file
                       else torch.randn(20, *grid, generator=gi)).float().to(device)]
    if is_vace:
        gi = torch.Generator(device="cpu").manual_seed(args.base_seed + 3)
        extra["vace_context"] = [(torch.load(args.vace_context_file, map_location="cpu") if args.vace_context_file
                                  else torch.randn(cfg["vace_in_dim"], *grid, generator=gi)).float().to(device)]
        extra["vace_context_scale"] = args.vace_context_scale
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    latent = mca.sample(model, noise, ctx, ctx_null, sampling_steps=args.sample_steps, shift=args.sample_shift,
                        guide_scale=args.sample_guide_scale, solver=args.sample_solver, layout=layout,
                        model_kwargs=extra, sigma_grid="upstream")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logging.info(f"denoising: {dt:.2f} s, {args.sample_steps / dt:.3f} steps/s")
    if rank == 0:
        out = args.save_file or f"{args.task}_{args.size.replace('*', 'x')}_{args.base_seed}_latent.pt"
        torch.save(latent.cpu(), out)
        logging.info(f"Saving the final latent to {out} (no VAE decoder in this repository)")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return latent


def _generate_ti2v(args, device, rank, world, layout):
    """--task ti2v-5B: MagCache4Wan2.2/magcache_generate.py:719-745 -- one dense 5B model on the 48-channel latent of the
    Wan2.2 VAE (stride 4 x 16 x 16).  Text-to-video by default; --y_file (torch tensor [48, 1, H/16, W/16]: the VAE latent
    of the conditioning image, which the absent VAE would produce from --image) switches to image-to-video, where the
    first latent frame is re-imposed every step and its tokens carry timestep 0 (per-token timesteps).  The reference
    picks the mag_ratios table by the same condition (:735-738)."""
    import torch
    import torch.distributed as dist
    from magcache_amd import model as M
    from magcache_amd import wan22
    from magcache_amd.engine import synthetic_weights
    assert world == 1, "ti2v-5B runs on one GPU in this repository"
    cfg = wan22.WAN22_TI2V_5B
    H, W = SIZE_CONFIGS[args.size][1], SIZE_CONFIGS[args.size][0]
    grid = ((args.frame_num - 1) // 4 + 1, H // 16, W // 16)
    logging.info(f"Generation job args: {args}")
    cls = type("WanModelHIP_TI2V", (M.WanModelHIP,), {})
    model = cls(cfg, grid, device=device, calibration=args.magcache_calibration)
    files = sorted(glob.glob(os.path.join(args.ckpt_dir or "", "*.safetensors")))
    if files:
        from safetensors.torch import load_file
        sd = {}
        for f in files:
            sd.update(load_file(f, device="cpu"))
        model.load_state_dict(sd)
    else:
        logging.warning("no *.safetensors under --ckpt_dir: seeded RANDOM-INIT weights of the architecture")
        model.engine.load_weights(synthetic_weights(cfg, seed=0, device=device))
    i2v = args.y_file is not None
    if args.use_magcache:
        table = wan22.table_without_pad("wan2.2_ti2v_5B_i2v" if i2v else "wan2.2_ti2v_5B_t2v")     # :735-738
        wan22.init_magcache(model, table, args.sample_steps, args.magcache_thresh, args.magcache_K, args.retention_ratio,
                            split_steps=None, mode="t2v")                                             # :739
    if args.magcache_calibration:
        wan22.init_magcache_calibration(model, args.sample_steps)                                    # :742
    prompt = args.prompt or EXAMPLE_PROMPT
    ctx = _context(args.context_file, prompt, args.base_seed, cfg["text_dim"], device)
    ctx_null = _context(args.context_null_file, "", args.base_seed + 1, cfg["text_dim"], device)
    g = torch.Generator(device=device).manual_seed(args.base_seed)
    noise = torch.randn(cfg["in_dim"], *grid, dtype=torch.float32, device=device, generator=g)
    z = torch.load(args.y_file, map_location="cpu").float().to(device) if i2v else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    latent = wan22.sample_ti2v(model, noise, ctx, ctx_null, sampling_steps=args.sample_steps, shift=args.sample_shift,
                               guide_scale=args.sample_guide_scale, z_first=z, solver=args.sample_solver,
                               sigma_grid="upstream")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logging.info(f"denoising: {dt:.2f} s, {args.sample_steps / dt:.3f} steps/s")
    out = args.save_file or f"{args.task}_{args.size.replace('*', 'x')}_{args.base_seed}_latent.pt"
    torch.save(latent.cpu(), out)
    logging.info(f"Saving the final latent to {out} (no VAE decoder in this repository)")
    return latent


def _generate_wan22(args, device, rank, world, layout):
    """--task t2v-A14B / i2v-A14B: the two-expert loop of MagCache4Wan2.2/magcache_generate.py (:683-800) on two
    engines.  --ckpt_dir follows the upstream layout (high_noise_model/ and low_noise_model/ with *.safetensors);
    i2v-A14B takes --y_file ([20, F, H/8, W/8]: mask + first-frame latent), there is no CLIP branch in Wan2.2."""
    import torch
    import torch.distributed as dist
    from magcache_amd import wan22
    from magcache_amd.engine import synthetic_weights
    d = wan22.WAN22_DEFAULTS[args.task]
    mode = "i2v" if args.task.startswith("i2v") else "t2v"
    cfg = wan22.WAN22_I2V_A14B if mode == "i2v" else wan22.WAN22_T2V_A14B
    H, W = SIZE_CONFIGS[args.size][1], SIZE_CONFIGS[args.size][0]
    grid = ((args.frame_num - 1) // 4 + 1, H // 8, W // 8)
    logging.info(f"Generation job args: {args}")
    sp = dict(sp_rank=layout.sp_rank, sp_size=layout.sp_size, sp_group=layout.sp_group) if layout else {}
    high, low = wan22.make_experts(cfg, grid, device=device, calibration=args.magcache_calibration, **sp)
    for seed, (m, sub) in enumerate(((high, "high_noise_model"), (low, "low_noise_model"))):
        files = sorted(glob.glob(os.path.join(args.ckpt_dir or "", sub, "*.safetensors")))
        if files:
            from safetensors.torch import load_file
            sd = {}
            for f in files:
                sd.update(load_file(f, device="cpu"))
            m.load_state_dict(sd)
            logging.info(f"{sub}: loaded {len(sd)} tensors")
        else:
            logging.warning(f"no {sub}/*.safetensors under --ckpt_dir: seeded RANDOM-INIT weights of the architecture")
            m.engine.load_weights(synthetic_weights(cfg, seed=seed, device=device))
    split = wan22.high_noise_steps(args.sample_shift, args.sample_steps, d["boundary"])          # :696-697
    if args.use_magcache:
        wan22.init_magcache(high, wan22.table_without_pad("wan2.2_i2v_A14B" if mode == "i2v" else "wan2.2_t2v_A14B"),
                            args.sample_steps, args.magcache_thresh, args.magcache_K, args.retention_ratio,
                            split_steps=split, mode=mode)                                          # :698, :774
    if args.magcache_calibration:
        wan22.init_magcache_calibration(high, args.sample_steps)                                   # :704, :780
    prompt = args.prompt or EXAMPLE_PROMPT
    ctx = _context(args.context_file, prompt, args.base_seed, cfg["text_dim"], device)
    ctx_null = _context(args.context_null_file, "", args.base_seed + 1, cfg["text_dim"], device)
    g = torch.Generator(device=device).manual_seed(args.base_seed)
    noise = torch.randn(16, *grid, dtype=torch.float32, device=device, generator=g)
    y = None
    if mode == "i2v":
        gi = torch.Generator(device="cpu").manual_seed(args.base_seed + 2)
        y = (torch.load(args.y_file, map_location="cpu") if args.y_file
             else torch.randn(20, *grid, generator=gi)).float().to(device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    latent = wan22.sample(high, low, noise, ctx, ctx_null, d["boundary"], sampling_steps=args.sample_steps,
                          shift=args.sample_shift, guide_scale=args.sample_guide_scale, y=y, solver=args.sample_solver,
                          layout=layout, sigma_grid="upstream")
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    logging.info(f"denoising: {dt:.2f} s, {args.sample_steps / dt:.3f} steps/s ({split} high-noise steps)")
    if rank == 0:
        out = args.save_file or f"{args.task}_{args.size.replace('*', 'x')}_{args.base_seed}_latent.pt"
        torch.save(latent.cpu(), out)
        logging.info(f"Saving the final latent to {out} (no VAE decoder in this repository)")
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return latent


if __name__ == "__main__":
    generate(_parse_args())
