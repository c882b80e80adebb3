#!/usr/bin/env python3
"""
bench.py — walker-logL evals/s on Pantheon+-shaped full-covariance chi^2 (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--walkers-per-gpu n] [--scaling weak|strong] [--n-sn 1701]

`python bench.py --gpus N` with N > 1 starts the N ranks itself (torch.distributed.run, one process per GPU, RCCL) from
a parent that never touches a GPU; the driver's own launch
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...
is recognised by RANK / WORLD_SIZE in the environment and runs the rank directly.

A "step" is one pass of the hot path over the ensemble: every rank evaluates log P for the walkers it owns with theta
already resident in HBM; for N > 1 the step first all-gathers the walker positions (the exchange an ensemble move needs
to pick partners from the complementary set).  Rank 0 prints ONE JSON line.

Workloads (BASELINE.json configs):
  N = 1, 2, 4, 8   configs[1]: Pantheon+ 1701-SN full-cov flat-LambdaCDM, 4096 walkers PER GPU at every N (weak scaling: one
                   per-GPU shape, so value_N / (N value_1) is one workload).
  --scaling strong configs[3]: 65536 walkers in total at every N (65536 / N per GPU; at N = 8 that is 8192 per GPU).
  --mode inprocess SURVEY 8e form (1): ONE host process, one handle over k devices (`--devices all` or a list of ordinals; an
                   ordinal may repeat: a rehearsal on a one-GPU box, labelled as such), host numpy buffers through cf_eval,
                   --walkers-total walkers per call (default 65536) -- how the reference's own scripts would run (sn/pantheon.py:119-125).
Data are synthetic (the real covariance is not in the reference snapshot): see cosmology-model-fit_amd/synthetic.py.
All arithmetic float64.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X FP64 matrix (= vector) datasheet peak: 256 CU x 4 SIMD x 2048 flop / 64 clk x 2.4 GHz

def flops_per_eval_solve(n):
    """Algorithmic FP64 flops of the chi^2 phase per walker: TRSV n(n-1) + n divisions + 2n norm."""
    return n * (n - 1) + 3 * n


def plan(mode, scaling, world, walkers_per_gpu=None, walkers_total=65536, devices="all"):
    """Walkers per GPU / in total, and the label of the run.  Pure arithmetic (tests/test_bench_cpu.py).
    ranks + weak: `walkers_per_gpu` (default 4096) on every rank at EVERY world size -- one per-GPU shape across N.
    ranks + strong: `walkers_total` over the ranks, whole 32-walker panels per rank.
    inprocess: `walkers_total` per call of ONE handle over `devices`."""
    if mode == "inprocess":
        if world != 1:
            raise ValueError("--mode inprocess is ONE process (do not start it under torch.distributed.run)")
        devs = "all" if devices == "all" else [int(x) for x in str(devices).split(",") if x != ""]
        if devs != "all" and (not devs or min(devs) < 0):
            raise ValueError("--devices must be 'all' or a comma-separated list of HIP ordinals")
        if walkers_total < 32:
            raise ValueError("--walkers-total must be at least one 32-walker panel")
        return {"mode": "inprocess", "devices": devs, "walkers_total": walkers_total, "walkers_per_gpu": None, "scaling": "strong",
                "rehearsal": devs != "all" and len(set(devs)) < len(devs)}
    if scaling == "strong":
        if walkers_total % (32 * world):
            raise ValueError("--walkers-total must be a multiple of 32 x the number of GPUs")
        wl = walkers_total // world
    else:
        wl = walkers_per_gpu or 4096
    return {"mode": "ranks", "devices": None, "walkers_total": wl * world, "walkers_per_gpu": wl, "scaling": scaling, "rehearsal": False}


def executed_mfmas_solve(n_sn, walkers, skip_padding=True):
    """v_mfma_f64_16x16x4_f64 instructions one launch of the inverse-GEMM solve executes.  Per 16-walker panel and 64-row block rb:
    4 tiles x 16 (rb + 1) K-steps; with `skip_padding` (the library's default) less the all-padding tiles of the last row
    block (16 n_rb K-steps each) and the zero tiles of every diagonal block (6 tile-groups of 4 K-steps; of the last block only those
    of its kept tiles)."""
    n_rb = (n_sn + 63) // 64
    per_panel = 64 * n_rb * (n_rb + 1) // 2
    if skip_padding:
        nt_last = max(1, min(4, ((n_sn + 15) // 16 * 16 - 64 * (n_rb - 1)) // 16))
        zero_groups = lambda nt: nt * (nt - 1) // 2  # tile j of a diagonal block skips the groups m > j: sum over kept tiles of (nt - 1 - j) ... = nt (nt - 1) / 2
        per_panel -= (4 - nt_last) * 16 * n_rb                      # padded tiles of the last row block
        per_panel -= 4 * ((n_rb - 1) * zero_groups(4) + (zero_groups(4) - sum(3 - j for j in range(nt_last, 4))))
    return ((walkers + 15) // 16) * per_panel


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--walkers-per-gpu", type=int, default=None, help="weak scaling: walkers per GPU, default 4096 (BASELINE configs[1]) at every N")
    ap.add_argument("--mode", default="ranks", choices=["ranks", "inprocess"],
                    help="ranks: one process per GPU (torch.distributed / RCCL), theta resident in HBM (the bench contract); inprocess: one "
                         "process, one handle over --devices, host buffers through cf_eval (SURVEY 8e form 1)")
    ap.add_argument("--devices", default="all", help='--mode inprocess: "all" or comma-separated HIP ordinals, e.g. 0,0,0,0 (rehearsal on one GPU)')
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: fixed walkers per GPU; strong: 65536 walkers in total (--walkers-total) at every N")
    ap.add_argument("--walkers-total", type=int, default=65536, help="ensemble size of --scaling strong")
    ap.add_argument("--fde", default="lcdm", choices=["lcdm", "cpl"],
                    help="desi_cmb_des5y only: dark energy as shipped (Lambda) or the script's commented w0waCDM line "
                         "(bao/desi_cmb_des5y.py:28-31), BASELINE configs[2] as worded")
    ap.add_argument("--n-sn", type=int, default=1701)
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="CPU-seconds of work of the cpu_baseline sample (rank 0, N=1)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precondition-ms", type=float, default=300.0,
                    help="untimed evaluations of the same workload BEFORE the W warm-up steps, until the device has been busy this "
                         "long: a GPU that has just left idle ramps its clock over ~0.1 s (profiles/r02_event_overhead.txt), and a "
                         "sampler runs for hours at the sustained clock.  0 = none")
    ap.add_argument("--solve", default="default", choices=["default", "blocked", "inverse"],
                    help="solve kernel: blocked TRSM or the inverse-GEMM form (default: the library's choice)")
    ap.add_argument("--workload", default="pantheon", choices=["pantheon", "desi_cmb_des5y", "desi_des5y_bbn_theta_star"],
                    help="pantheon = BASELINE configs[1] (the headline, default); desi_cmb_des5y = configs[2] shape "
                         "(N=1820 SN + 14 BAO + Planck/ACT CMB, physical-density E(z)); desi_des5y_bbn_theta_star = "
                         "configs[4] shape (N=1820 SN + 13 BAO + l_A + BBN prior, thawing dark energy, log P as nautilus "
                         "would batch it); both on the committed fixture data")
    args = ap.parse_args()

    if args.mode == "inprocess":
        return run_inprocess(args)
    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N`: this parent starts one child per GPU and never initialises a GPU itself
        import socket
        import subprocess

        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd).returncode)

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    args.gpus = world
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to time instead")
    pkg = importlib.import_module("cosmology-model-fit_amd")
    if not os.path.exists(pkg._lib.LIB_PATH) and local_rank == 0:
        pkg.build()  # normally built by __graft_entry__.build(); the Makefile moves the finished file into place atomically
    n_visible = torch.cuda.device_count()
    backend = os.environ.get("BENCH_DIST_BACKEND", "nccl")  # "gloo": rehearsal only (ranks sharing one GPU)
    if world > n_visible and backend == "nccl":
        sys.exit(f"bench.py: {world} ranks but {n_visible} GPU(s) visible; RCCL needs one GPU per rank "
                 "(BENCH_DIST_BACKEND=gloo rehearses several ranks on one GPU)")
    local_rank = local_rank % n_visible
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # under torch.distributed.run (RANK in the environment) the rank joins the process group and every step carries the
    # all-gather of positions -- also with ONE rank, so that `torchrun --nproc-per-node 1 bench.py` exercises RCCL
    # initialisation and the device collective on a one-GPU box; plain `python bench.py` has no process group
    use_dist = "RANK" in os.environ
    if use_dist:
        import torch.distributed as dist
        # RCCL and gloo print banners ("RCCL version : ...", "[Gloo] Rank 0 is connected ...") on the process's stdout when the
        # communicator comes up; stdout carries ONE JSON line, so fd 1 points at stderr until the first collective has run
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=dev)
            else:
                dist.init_process_group(backend=backend)
            dist.barrier()  # rank 0 may have been building the library: nobody loads it before this point
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)  # the first device collective creates the communicator
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    sn = pkg.sn_pantheon
    try:
        shape = plan("ranks", args.scaling, world, args.walkers_per_gpu, args.walkers_total)
    except ValueError as e:
        sys.exit(str(e))
    Wl, W_total = shape["walkers_per_gpu"], shape["walkers_total"]

    solve_kw = {} if args.solve == "default" else {"solve": args.solve}
    if args.workload == "desi_cmb_des5y":
        # joint likelihood of bao/desi_cmb_des5y.py: real DES-Dovekie redshifts + real DESI FS+Lya BAO data from the
        # golden fixture (tests/golden/bao_desi_cmb_des5y.npz), seeded synthetic SN covariance
        g = np.load(os.path.join(ROOT, "tests", "golden", "bao_desi_cmb_des5y.npz"))
        rng = np.random.default_rng(0)
        A = 0.01 * rng.standard_normal((g["sigma"].size, 40))
        chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
        lk = pkg.likelihoods.DesiCmbDes5y(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"],
                                          g["bao_inv_cov"], chol=chol, device=local_rank, fde=args.fde, **solve_kw)
        box = [(-0.5, 0.5), (60.0, 75.0), (0.010, 0.030), (0.01, 0.25), (-4.5, 4.5)]  # bao/desi_cmb_des5y.py:156-161
        if args.fde == "cpl":
            box += [(-3.0, 1.0), (-3.0, 2.0)]  # w0, wa: bao/desi_fs_lya_cmb.py:135-136
        box = np.array(box)
        args.n_sn, ndim, kind = int(g["z_cmb"].size), len(box), pkg.CF_OUT_LOGL
        syn = dict(g=g, chol=chol)
    elif args.workload == "desi_des5y_bbn_theta_star":
        g = np.load(os.path.join(ROOT, "tests", "golden", "bao_desi_des5y_bbn_theta_star.npz"))
        rng = np.random.default_rng(0)
        A = 0.01 * rng.standard_normal((g["sigma"].size, 40))
        chol = np.linalg.cholesky(np.diag(g["sigma"] ** 2) + A @ A.T)
        lk = pkg.likelihoods.DesiDes5yBbnThetaStar(g["z_cmb"], g["z_hel"], g["obs"], None, g["bao_z"], g["bao_val"], g["bao_qty"],
                                                   g["bao_inv_cov"], chol=chol, device=local_rank)
        box = g["bounds"]  # bao/desi_des5y_bbn_theta_star.py:122-130
        args.n_sn, ndim, kind = int(g["z_cmb"].size), 5, pkg.CF_OUT_LOGP
        syn = dict(g=g, chol=chol)
    else:
        syn = pkg.synthetic.pantheon_like(n_sn=args.n_sn, seed=0)
        lk = sn.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"], device=local_rank, **solve_kw)
        box, ndim, kind = sn.bounds, 4, pkg.CF_OUT_LOGP
    eng = lk.engine

    theta_all_host = pkg.synthetic.walkers(box, W_total, seed=0)
    mine = slice(rank * Wl, (rank + 1) * Wl)
    theta_local = torch.from_numpy(theta_all_host[mine].copy()).to(dev)
    theta_all = torch.empty((W_total, ndim), dtype=torch.float64, device=dev)
    logp = torch.empty(Wl, dtype=torch.float64, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        if use_dist:
            # exchange step of an ensemble move: every rank needs the complementary walkers' positions
            if backend == "nccl":
                dist.all_gather_into_tensor(theta_all, theta_local)
            else:  # rehearsal backend without device collectives: stage through the host
                host = torch.empty((W_total, ndim), dtype=torch.float64)
                dist.all_gather_into_tensor(host, theta_local.cpu())
                theta_all.copy_(host)
        eng.eval_device(theta_local.data_ptr(), Wl, logp.data_ptr(), kind, stream)

    def fence():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_pass():
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize fences; seconds of the K steps."""
        for _ in range(args.warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        return time.perf_counter() - t0

    # Two passes of the same W + K protocol.  FROM IDLE: the GPU has done nothing but set-up copies so far and its clock is
    # still ramping (reported as `from_idle`, never as `value`).  Then `--precondition-ms` of the same evaluations, untimed,
    # and the pass that counts: the sustained clock, which is what a sampler running for hours sees.
    n_pre, dt_idle = 0, None
    if args.precondition_ms > 0:
        dt_idle = timed_pass()
        t_pre = time.perf_counter()
        while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
            for _ in range(16):
                eng.eval_device(theta_local.data_ptr(), Wl, logp.data_ptr(), kind, stream)
            torch.cuda.synchronize()
            n_pre += 16
    # kernel durations from HIP events on the stream the kernels run on, SAMPLED over the timed region: recording them on
    # every step costs 5 % of the step (profiles/r02_event_overhead.txt); the first warm-up step is sampled too and dropped
    stride = max(1, min(8, args.steps // 5))
    eng.enable_timing(min(args.steps + args.warmup, 4096), stride)
    dt = timed_pass()
    kms = eng.kernel_ms3()
    n_warm_samples = (args.warmup + stride - 1) // stride  # samples that fell into the W warm-up steps
    kms = kms[n_warm_samples:] or kms
    eng.enable_timing(0)
    dt_by_rank, allgather_ms = [dt], None
    if use_dist:
        # every rank's own clock around the K steps (value uses the MAX), and the exchange alone: K all-gathers with nothing else
        own = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        per_rank = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(per_rank, own)
        dt_by_rank = [float(x.item()) for x in per_rank]
        t = torch.tensor([dt, dt_idle or 0.0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, dt_idle = float(t[0].item()), (float(t[1].item()) if dt_idle is not None else None)
        got = theta_all.cpu().numpy()
        assert np.array_equal(got, theta_all_host), "all-gather of walker positions is wrong"
        if backend == "nccl":
            fence()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                dist.all_gather_into_tensor(theta_all, theta_local)
            fence()
            ag = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
            dist.all_reduce(ag, op=dist.ReduceOp.MAX)
            allgather_ms = float(ag.item()) / args.steps * 1e3

    result = logp.cpu().numpy()
    assert np.all(np.isfinite(result)), "in-box walkers must give a finite log-probability"

    # which physical device each rank ran on (uuid / PCI address), gathered so that rank 0 can show they are distinct
    props = torch.cuda.get_device_properties(dev)
    ident = "%s pci %04x:%02x:%02x uuid %s" % (props.name, getattr(props, "pci_domain_id", 0), getattr(props, "pci_bus_id", 0),
                                              getattr(props, "pci_device_id", 0), getattr(props, "uuid", "?"))
    idents = [ident]
    if use_dist:
        idents = [None] * world
        dist.all_gather_object(idents, ident)

    if rank == 0:
        solve_kernel = "tri_gemm_chi2_kernel" if eng.info()["solve_mode"] == pkg.CF_SOLVE_INVERSE_GEMM else "trsm_chi2_kernel"
        walker_ms = float(np.mean([k[0] for k in kms]))
        blocks_ms = float(np.mean([k[1] for k in kms]))
        solve_ms = float(np.mean([k[2] for k in kms]))
        solve_flops = flops_per_eval_solve(args.n_sn) * Wl
        achieved = solve_flops / (solve_ms * 1e-3) / 1e12
        traffic, traffic_source = pmc_traffic(args.workload if args.fde == "lcdm" else f"{args.workload}:{args.fde}", args.n_sn, Wl,
                                              solve_kernel)
        out = {
            # `value`: theta already resident in HBM when the timed region starts (the bench contract); the same metric as
            # SURVEY 8(d) words it -- host buffers in and out, PCIe and the synchronisation included -- is `value_host_visible`
            "metric": {"pantheon": "walker-logL evals/s, Pantheon+ 1701-SN full-cov chi2, theta resident in HBM",
                       "desi_cmb_des5y": "walker-logL evals/s, DESI BAO + Planck/ACT CMB + DES-SN joint chi2 (config 3 shape), theta resident in HBM",
                       "desi_des5y_bbn_theta_star": "walker-logL evals/s, DESI BAO + l_A + BBN + DES-SN joint log P (config 5 shape), theta resident in HBM",
                       }[args.workload],
            "value": W_total * args.steps / dt,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_by_rank": [x / args.steps * 1e3 for x in dt_by_rank],
            "allgather_ms_per_step": allgather_ms,  # K all-gathers of the positions alone, max over ranks (None: no RCCL in this run)
            "mode": "ranks",
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if args.workload == "pantheon" else
                    "fixture redshifts / magnitudes / BAO data (tests/golden) + seeded synthetic SN covariance, synthetic walkers",
            "config": {
                "workload": (f"Pantheon+-shaped {args.n_sn}-SN full-cov flat-LCDM chi2 + prior, "
                             f"{Wl} walkers per GPU per step (BASELINE configs[1] at N=1), G=4000, theta resident in HBM")
                if args.workload == "pantheon" else
                (f"bao/desi_cmb_des5y.py joint log L: {args.n_sn} SNe (velocity step) + 14 BAO (PCHIP D_H, F_AP) + Planck/ACT "
                 f"(R, l_A, wb), physical-density E(z), {Wl} walkers per GPU per step") if args.workload == "desi_cmb_des5y" else
                (f"bao/desi_des5y_bbn_theta_star.py joint log P: {args.n_sn} SNe + 13 BAO (exact D_H) + l_A + BBN prior, "
                 f"physical-density E(z) with thawing dark energy, {Wl} walkers per GPU per step"),
                "workload_key": args.workload if args.fde == "lcdm" else f"{args.workload}:{args.fde}",
                "walkers_per_gpu": Wl, "walkers_total": W_total, "n_sn": args.n_sn, "n_grid": 4000, "ndim": ndim,
                "parallelism": f"walkers sharded over {world} GPU(s)" + (
                    "" if not use_dist else
                    ", RCCL all-gather of positions per step" if backend == "nccl" else
                    f", {backend} all-gather of positions staged through the host (rehearsal: not RCCL)"),
            },
            "backend": ("rccl (torch.distributed nccl)" if backend == "nccl" else backend) if use_dist else None,
            "world_size": world,
            "device_ids": idents,
            "distinct_devices": len(set(idents)),
            "roofline": {
                "kernel": solve_kernel,
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "flops_per_launch": solve_flops,
                "avg_kernel_ms": solve_ms,
            },
            "kernels_ms": {"walker_kernel": walker_ms, "small_blocks_kernel": blocks_ms if (args.workload != "pantheon") else None,
                           solve_kernel: solve_ms},
            "kernel_timing": f"HIP events on the launch stream, every {stride}th of the {args.steps} timed steps ({len(kms)} samples); a "
                             "sampled step pays for its four event records (profiles/r02_event_overhead.txt), so kernels_ms overstate the "
                             "untimed steps' kernels by 0.5-2.5 % and may sum to more than ms_per_step; roofline.frac is that much conservative",
            "preconditioning": {"untimed_evaluations_before_warmup": n_pre, "ms": args.precondition_ms},
            "from_idle": None if dt_idle is None else {
                "value": W_total * args.steps / dt_idle, "ms_per_step": dt_idle / args.steps * 1e3,
                "note": "the same W warm-up + K timed steps run FIRST, on a GPU straight from idle (clock still ramping); "
                        "`value` is the second pass, after the preconditioning"},
            "device": {k: eng.info()[k] for k in ("gcn_arch", "cu_count")},
        }
        # the HBM view next to the matrix-core view.  `hbm_gbps_measured` = PMC bytes of the solve
        # kernel / its duration (small by design: the factor is reused by every walker from L2 / Infinity Cache);
        # `trsv_equivalent_gbps` = what a design without reuse would have to stream (8 N (N+1) / 2 bytes per eval),
        # quoted for comparison with the CPU path only -- it exceeds the 8 TB/s HBM peak because of the reuse.
        if solve_kernel == "tri_gemm_chi2_kernel":
            # frac = mfma_busy x useful / executed x clock / 2.4 GHz: the matrix instructions the launch EXECUTES are arithmetic
            # (diagonal blocks and padded rows in full), their 64 cycles each on 1024 SIMDs against the kernel's duration at the
            # clock the chip holds under it (in-kernel s_memtime / s_memrealtime of the diagnostic build, replayed from profiles/)
            n_mfma = executed_mfmas_solve(args.n_sn, Wl)
            clock_ghz, clock_source, bare = replay_clock()
            r = out["roofline"]
            r["executed_tflops"] = n_mfma * 2048 / (solve_ms * 1e-3) / 1e12
            r["useful_over_executed"] = solve_flops / (n_mfma * 2048.0)
            r["clock_ghz"], r["clock_source"] = clock_ghz, clock_source
            r["mfma_busy"] = n_mfma * 64 / 1024.0 / (solve_ms * 1e-3 * clock_ghz * 1e9) if clock_ghz else None
            r["frac_factors"] = None if not clock_ghz else {"mfma_busy": r["mfma_busy"], "useful_over_executed": r["useful_over_executed"],
                                                            "clock_over_2p4": clock_ghz / 2.4}
            r["bare_mfma_loop_tflops"] = bare  # what a register-only v_mfma_f64_16x16x4_f64 loop sustains on this silicon (same source)
        out["roofline"]["hbm_gbps_measured"] = traffic / (solve_ms * 1e-3) / 1e9 if traffic else None
        out["roofline"]["trsv_equivalent_gbps"] = 8.0 * args.n_sn * (args.n_sn + 1) / 2 * out["value"] / 1e9
        if world == 1 and not args.no_cpu_baseline:  # context probes (this and cpu_baseline) stay out of profiled runs
            out["roofline"]["hbm_stream_triad_gbps"] = stream_triad_gbps(torch, dev)
        if world == 1 and args.workload == "pantheon":
            # the ctypes boundary as emcee / nautilus call it: host numpy in, host numpy out (PCIe + sync included).
            # Reported for DESIGN.md; never the headline `value`.
            th_host = theta_all_host[mine]
            for _ in range(3):
                lk.log_probs_vectorized(th_host)
            t0 = time.perf_counter()
            reps = max(50, args.steps)  # ~15 ms: a handful of calls is at the mercy of one scheduling hiccup
            for _ in range(reps):
                host_res = lk.log_probs_vectorized(th_host)
            # SURVEY 8(d)'s own wording of the metric: wall-clock, host-visible, H2D of theta and D2H of the results
            # included -- first-class beside `value` (which is theta-resident, as the bench contract asks)
            out["value_host_visible"] = Wl * reps / (time.perf_counter() - t0)
            out["host_visible"] = {"metric": "walker-logL evals/s, Pantheon+ 1701-SN full-cov chi2, host numpy buffers through cf_eval "
                                             "(H2D + kernels + D2H + sync: SURVEY 8d's wording of the metric)",
                                   "value": out["value_host_visible"], "unit": "evals/s", "calls": reps}
            assert np.array_equal(host_res, result)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(pkg, args, syn, lk, box, kind, theta_all_host, result, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


def stream_triad_gbps(torch, dev, n=1 << 27, reps=10):
    """What this device's HBM delivers on a STREAM triad a = b + s * c (3 x 1 GiB of float64), after the timed region."""
    b = torch.ones(n, dtype=torch.float64, device=dev)
    c = torch.ones(n, dtype=torch.float64, device=dev)
    a = torch.empty_like(b)
    torch.add(b, c, alpha=0.5, out=a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        torch.add(b, c, alpha=0.5, out=a)
    e1.record()
    torch.cuda.synchronize()
    return 3 * 8 * n * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def host_cpu():
    """CPU model and logical CPU count of the box the baseline ran on."""
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"model": model, "logical_cpus": os.cpu_count(), "usable": len(os.sched_getaffinity(0))}


def replay_clock():
    """(in-kernel shader clock of the solve kernel in GHz, source file, bare-MFMA-loop TFLOP/s) from the newest committed
    profiles/r*_solve_clock.json (tools/solve_clock.py with the diagnostic build + tools/coexec_f64_rate on the same box); Nones
    when there is none.  The production kernel executes no stamp, so the number is replayed and labelled as such."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_solve_clock.json")), reverse=True):
        try:
            prof = json.load(open(path))
            return float(prof["solve_kernel_clock_ghz_median"]), os.path.relpath(path, ROOT), prof.get("bare_mfma_loop_tflops")
        except Exception:
            continue
    return None, None, None


def run_inprocess(args):
    """SURVEY 8e form (1): the sampler's host process holds ONE handle over k devices and calls the vectorised callback with
    host numpy buffers; cf_eval cuts the rows into contiguous slices of whole panels, one host thread + stream per device."""
    import torch

    if "RANK" in os.environ:
        sys.exit("--mode inprocess is ONE process: do not start it under torch.distributed.run")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (no HIP device visible); there is no CPU path to time instead")
    if args.workload != "pantheon":
        sys.exit("--mode inprocess times the headline workload (pantheon)")
    try:
        shape = plan("inprocess", args.scaling, 1, None, args.walkers_total, args.devices)
    except ValueError as e:
        sys.exit(str(e))
    pkg = importlib.import_module("cosmology-model-fit_amd")
    sn = pkg.sn_pantheon
    syn = pkg.synthetic.pantheon_like(n_sn=args.n_sn, seed=0)
    lk = sn.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"], devices=shape["devices"])
    info = lk.engine.info()
    devs = list(info["devices"])
    W = shape["walkers_total"]
    theta = pkg.synthetic.walkers(sn.bounds, W, seed=0)
    t_pre = time.perf_counter()
    while (time.perf_counter() - t_pre) * 1e3 < args.precondition_ms:
        lk.log_probs_vectorized(theta)
    for _ in range(args.warmup):
        lk.log_probs_vectorized(theta)
    cpu0, t0 = time.process_time(), time.perf_counter()
    for _ in range(args.steps):
        res = lk.log_probs_vectorized(theta)
    dt, cpu = time.perf_counter() - t0, time.process_time() - cpu0
    assert np.all(np.isfinite(res))
    one = sn.PantheonLikelihood(syn["z_cmb"], syn["z_hel"], syn["obs"], chol=syn["chol"], device=devs[0])
    assert np.array_equal(one.log_probs_vectorized(theta[:4096]), res[:4096]), "replicas changed a walker's result"
    solve_flops = flops_per_eval_solve(args.n_sn) * W
    idents = []
    for d in sorted(set(devs)):
        pr = torch.cuda.get_device_properties(d)
        idents.append("%s pci %04x:%02x:%02x uuid %s" % (pr.name, getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0),
                                                          getattr(pr, "pci_device_id", 0), getattr(pr, "uuid", "?")))
    n_phys = len(set(devs))
    out = {
        "metric": "walker-logL evals/s, Pantheon+ 1701-SN full-cov chi2, host numpy buffers through ONE handle over k devices (cf_eval)",
        "value": W * args.steps / dt, "unit": "evals/s", "n_gpus": n_phys, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic", "mode": "inprocess",
        "config": {"workload": f"Pantheon+-shaped {args.n_sn}-SN full-cov flat-LCDM chi2 + prior, {W} walkers per call of ONE handle over "
                               f"{len(devs)} replica(s) on {n_phys} physical GPU(s) (SURVEY 8e form 1: sn/pantheon.py:119-125's dispatch as one batched call)",
                   "walkers_total": W, "n_sn": args.n_sn, "n_grid": 4000, "ndim": 4, "replicas": devs,
                   "parallelism": f"rows of theta split over {len(devs)} replicas (cf_split_rows), one host thread + stream each, no collective",
                   "rehearsal": "ordinals repeat: the replicas SHARE a GPU -- the split / threads / streams are exercised, the rate is not a k-GPU rate"
                   if shape["rehearsal"] else None},
        "device_ids": idents, "distinct_devices": n_phys,
        "host_cpu_seconds_per_call": cpu / args.steps,  # all host threads of this process (the waiting policy: DESIGN section 6)
        "roofline": {"kernel": "tri_gemm_chi2_kernel (whole call: H2D + kernels + D2H + sync on every replica)", "bound": "mfma",
                     "achieved": solve_flops / (dt / args.steps) / 1e12, "peak": FP64_MFMA_PEAK_TFLOPS * n_phys, "unit": "TFLOP/s",
                     "frac": solve_flops / (dt / args.steps) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * n_phys), "traffic": None},
    }
    print(json.dumps(out), flush=True)
    lk.engine.close()
    one.engine.close()


def pmc_traffic(workload, n_sn, walkers, kernel):
    """(HBM bytes per launch of the solve kernel, source file) from the newest committed PMC profile of THIS configuration
    (profiles/r*_pmc_traffic*.json, written by tools/pmc_profile.sh + tools/pmc_summary.py: separate --pmc passes of this
    same bench command, gfx950 FETCH_SIZE correction); (None, None) when there is none.  Counters cannot be read from
    inside the timed run, so the number is replayed from that file and labelled as such (`traffic_source`)."""
    import glob

    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic*.json")), reverse=True):
        try:
            prof = json.load(open(path))
            cfg = prof["config"]
            if cfg["n_sn"] == n_sn and cfg["walkers_per_gpu"] == walkers and cfg.get("workload", "pantheon") == workload \
                    and kernel in prof["kernels"]:
                return prof["kernels"][kernel]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except Exception:
            continue
    return None, None


def oracle_likelihood(pkg, args, syn, lk, box):
    """The oracle's statement (oracle/oracle_np.Likelihood, evaluated by the C restatement) of the likelihood being timed."""
    from oracle import oracle_np as onp

    if args.workload == "pantheon":
        sn = pkg.sn_pantheon
        return onp.Likelihood(ndim=4, z_max=lk.z_max, offset=onp.Slot(0), H0=onp.Slot(1), Om=onp.Slot(2), v=onp.Slot(3),
                              z_cmb=syn["z_cmb"], z_hel=syn["z_hel"], obs=syn["obs"], chol=syn["chol"], bounds=sn.bounds,
                              gauss=[sn.H0_PRIOR])
    g, chol, d = syn["g"], syn["chol"], pkg.cmb_data.PLANCK_ACT
    phys = {k: d[k] for k in ("or_h2", "omnu_h2", "o_gamma_h2", "nu_m0", "nu_rho0", "nu_qs_sq", "nu_ws")}
    common = dict(z_max=lk.z_max, ez_model=onp.EZ_PHYSICAL, offset=onp.Slot(0), H0=onp.Slot(1), obh2=onp.Slot(2), och2=onp.Slot(3),
                  z_cmb=g["z_cmb"], z_hel=g["z_hel"], obs=g["obs"], chol=chol, bao_z=g["bao_z"], bao_val=g["bao_val"],
                  bao_qty=g["bao_qty"], bao_inv_cov=g["bao_inv_cov"], rd_fit=d["rd_fit"], cmb_prior=d["cmb_prior"],
                  zstar_fit=d["zstar_fit"], **phys)
    if args.workload == "desi_cmb_des5y":  # bao/desi_cmb_des5y.py:26-141 (tests/test_oracle_golden.py, test_variants.py)
        extra = dict(w0=onp.Slot(5), wa=onp.Slot(6)) if args.fde == "cpl" else {}
        return onp.Likelihood(ndim=len(box), fde=onp.FDE_CPL if args.fde == "cpl" else onp.FDE_LCDM, v=onp.Slot(4), z_turn=0.10563,
                              cmb_mode=1, cmb_inv_cov=d["cmb_inv_cov"], **extra, **common)
    inv = np.zeros((3, 3))  # bao/desi_des5y_bbn_theta_star.py:104-130: the l_A term alone, BBN prior, thawing dark energy
    inv[1, 1] = 1.0 / d["cmb_cov"][1, 1]
    return onp.Likelihood(ndim=5, fde=onp.FDE_THAWING, w0=onp.Slot(4), has_vstep=False, bao_dh_exact=True, cmb_mode=2,
                          cmb_inv_cov=inv, bounds=g["bounds"], gauss=[(2, float(g["bbn"][0]), float(g["bbn"][1]))], **common)


def cpu_baseline(pkg, args, syn, lk, box, kind, theta, gpu_out, budget_s):
    """The C restatement of the reference algorithm (oracle/, kind 'port') timed on this host's cores on a
    bounded sample of the same walkers; also reports the GPU-vs-CPU parity on that sample."""
    from oracle import oracle_c

    co = oracle_c.COracle(oracle_likelihood(pkg, args, syn, lk, box))
    usable = len(os.sched_getaffinity(0))
    # single-thread rate first (also sizes the sample)
    n1 = min(256, len(theta))
    t0 = time.perf_counter()
    co.eval(theta[:n1], kind, nthreads=1)
    dt1 = time.perf_counter() - t0
    rate1 = n1 / dt1
    # bounded sample: about `budget_s` CPU-seconds of work per leg (wall time = that / threads).
    # The first len(theta) walkers ARE the GPU batch (same generator stream), the rest are more draws
    # from the same prior box.
    n = int(min(65536, max(len(theta), rate1 * budget_s)))
    sample = pkg.synthetic.walkers(box, n, seed=0)
    m = len(theta)
    assert np.array_equal(sample[:m], theta)
    # every usable hardware thread, set explicitly -- and half of them (one per physical core on an SMT-2 host): the row-by-row
    # substitution streams the 11.6 MB factor per walker and two SMT siblings share a core's cache and load ports, so the
    # all-threads leg is not necessarily the faster one (256 threads: 1.25e4 evals/s, 128: 2.0e4 on the 2 x EPYC 9575F boxes
    # of this pool).  `value` is the better of the two, both are reported.
    legs = {}
    for nthreads in sorted({usable, max(1, usable // 2)}, reverse=True):
        co.eval(sample[:4 * nthreads], kind, nthreads=nthreads)  # spin the thread pool up
        t0 = time.perf_counter()
        ref = co.eval(sample, kind, nthreads=nthreads)
        legs[co.threads_used] = n / (time.perf_counter() - t0)
    cores = max(legs, key=legs.get)
    dt = n / legs[cores]
    rel = float(np.max(np.abs(gpu_out[:m] - ref[:m]) / np.abs(ref[:m])))
    return {
        "value": n / dt, "unit": "evals/s", "cores": cores, "kind": "port",
        "sample": f"{n} walkers from the same prior box (the first {m} are the timed GPU batch), C restatement "
                  f"oracle/cosmofit_oracle.c (-O2, no fast-math, OpenMP over walkers; timed with {sorted(legs)} threads of {usable} usable, "
                  f"`value` = the faster leg), ~{n / rate1:.0f} CPU-seconds of work per leg; single-thread: {n1 / dt1:.1f} evals/s on {n1} walkers",
        "single_thread_value": n1 / dt1,
        "evals_per_s_by_threads": {str(k): v for k, v in sorted(legs.items())},
        "parity_max_rel": rel,
        "parity_quantity": {0: "chi2", 1: "log L", 2: "log P"}[int(kind)],
        "host": host_cpu(),
    }


if __name__ == "__main__":
    main()
