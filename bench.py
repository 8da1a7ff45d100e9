#!/usr/bin/env python3
"""bench.py — Mpixels/s of the deferred-lit + atmosphere frame on MI355X.

A step is ONE frame of the hot path over a G-buffer that is already resident in HBM:
    lights pass (deferred/lights.comp)           -> scene colour
    transmittance LUT + sky-view LUT + composite (atmosphere/*.comp) -> scene colour
(the four dispatches the reference records every frame: renderer.cpp:383-415). Nothing is
cached between steps: both LUTs are recomputed each frame exactly like the reference
(skyview.cpp:799-893).

Workloads (BASELINE.json configs):
    c3  3840x2160, 64 spot lights, 1 GPU                      <- default at --gpus 1
    c4  7680x4320, 64 spot lights, rows cyclically tiled over --gpus N ranks; sky-view LUT rows
        sharded over the ranks + one all-gather; one RCCL gather of the RGBA16 tiles to rank 0 +
        one compose kernel                                       <- default at --gpus N > 1
    c2  1920x1080, 0 spot lights (sun by the composite, moon by the lights pass)
    c5  3840x2160, 256 spot lights per GPU, independent replicas, no collective

Prints ONE JSON line on rank 0 (see the repo contract), with `roofline` for the dominant
kernel of the run (its longest pass) and `cpu_baseline` (the scalar oracle timed on this box's
cores: all threads, and one thread). `python bench.py --gpus N` without a launcher starts its own N ranks.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

WORKLOADS = {
    "c2": dict(width=1920, height=1080, spots=0, tiled=False, sun_shadow=2048,
               label="C2 1920x1080 sun (composite, one 2048^2 analytic sun shadow map) + moon (lights pass), 0 spots"),
    "c3": dict(width=3840, height=2160, spots=64, tiled=False,
               label="C3 3840x2160 full atmosphere (LUTs 512x128 + 2048x1024 recomputed per frame) + 64 spot lights"),
    "c4": dict(width=7680, height=4320, spots=64, tiled=True,
               label="C4 7680x4320 deferred+atmosphere, cyclic row tiles over the ranks, RCCL gather + compose"),
    "c5": dict(width=3840, height=2160, spots=256, tiled=False, label="C5 3840x2160 per GPU, 256 spot lights, independent views"),
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
BLOCK_ROWS = 8  # row-tile block height (one kernel workgroup row)


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="auto", choices=["auto"] + sorted(WORKLOADS))
    ap.add_argument("--elevation", type=float, default=35.0, help="sun elevation in degrees")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-tiled", action="store_true",
                    help="testing: run the row-tiled code path (collectives, compose) even with a single rank")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (the product); gloo = host-staged rehearsal of the N-rank path")
    ap.add_argument("--same-device", action="store_true",
                    help="rehearsal: every rank renders on cuda:0 (several ranks on a one-GPU box; needs --backend gloo)")
    ap.add_argument("--cpu-rows", type=int, default=540, help="rows of the frame the all-threads CPU baseline shades")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements that are not part of `value` (profiling runs)")
    args = ap.parse_args()

    # The contract is ONE JSON line on stdout. Libraries write there too (RCCL prints its version banner to stdout on
    # this image), so everything else is sent to stderr and the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    # (dmabuf IPC is the only kind this pool's host driver supports; RCCL needs it before the first HIP call of a rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"WORLD_SIZE {world} != --gpus {args.gpus}")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher: this process has not touched the GPU yet (importing torch does
        # not), so it may start the N ranks itself - as CHILD processes, never by exec - relay rank 0's JSON line and
        # leave with their status.
        sys.exit(self_launch(args.gpus, real_stdout))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    if args.same_device:
        if args.backend != "gloo" and not os.environ.get("SZG_RCCL_LIBRARY"):
            # (tests/cpp/mock_rccl.cpp, named by SZG_RCCL_LIBRARY, stands in for RCCL in the one-GPU rehearsal of the C-ABI
            # collectives: it stages through host memory and does not mind ranks sharing a device)
            raise SystemExit("--same-device needs --backend gloo (RCCL refuses two ranks on one GPU)")
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_tiled:
        # Control plane (rendezvous, barrier, the max-over-ranks reduction of the wall time): gloo over 127.0.0.1.
        # Data plane with --backend nccl: RCCL behind the C-ABI (rowtile.Comm = szg_rowtile_comm, include/szg/abi.h).
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    import __graft_entry__ as entry

    entry.build(only_if_missing=True)
    from syzygy_amd import abi, lib, pipelines as pl, rowtile, scene

    name = args.workload if args.workload != "auto" else ("c3" if args.gpus == 1 else "c4")
    wl = WORKLOADS[name]
    W, H, SPOTS = wl["width"], wl["height"], wl["spots"]
    tiled = wl["tiled"] and (world > 1 or args.force_tiled)
    nranks = world if tiled else 1
    tile = rowtile.make_tile(H, rank, nranks, BLOCK_ROWS)
    rows = tile.local_rows if tile is not None else H
    stride_rows = rowtile.stride_rows(H, nranks, BLOCK_ROWS)

    # ---- inputs (seeded, synthetic: SURVEY 8d) ---------------------------------------
    syn = scene.SyntheticScene()
    atmosphere = scene.default_atmosphere(scene.sun_euler_for_elevation(args.elevation))
    atm, sun, moon = scene.atmosphere_baked(atmosphere, syn.bounds)
    cam = scene.camera_packed(scene.default_camera(), W / H)
    spots = scene.spot_ring(SPOTS)
    dev = f"cuda:{local_rank}"
    cameras = pl.TStagedBuffer(abi.CameraPacked, 1, dev)
    atmospheres = pl.TStagedBuffer(abi.AtmospherePacked, 1, dev)
    lights = pl.TStagedBuffer(abi.DirectionalLightPacked, 2, dev)
    cameras.push(cam)
    atmospheres.push(atm)
    lights.push([sun, moon])
    for b in (cameras, atmospheres, lights):
        b.recordCopyToDevice()

    # Two frames in flight like the reference (framebuffer.cpp:134): frame k renders into targets[k % 2] while the
    # tile gather of frame k-1 is still on the wire.
    targets = [pl.SceneTexture(W, max(stride_rows, 1), dev) for _ in range(2 if tiled else 1)]
    target = targets[0]
    sun_shadow = wl.get("sun_shadow", 0)
    deferred = pl.DeferredShadingPipeline((W, max(rows, 1)), max_spot_lights=max(SPOTS, 1), max_shadow_maps=1 if sun_shadow else 0,
                                          shadow_map_dim=sun_shadow, device_index=local_rank)
    sky = pl.SkyViewComputePipeline.create(device_index=local_rank)
    assert sky is not None
    rect = pl.rect(W, H)
    # G-buffer fill and shadow-map generation: producers of the path's inputs, outside the timed region
    for t in targets:
        deferred.recordGBufferFill(None, rect, t, 0, cameras, syn.fill, tile=tile)
    if sun_shadow:
        deferred.recordShadowMaps(None, lights, None, syn.fill)
    torch.cuda.synchronize()
    geometry_px_local = int((target.depth[:rows] > 0).sum().item())

    gathered = composed = None
    if tiled and rank == 0:
        gathered = [torch.empty((nranks, stride_rows, W, 4), dtype=torch.int16, device=dev) for _ in range(2)]
        composed = torch.empty((H, W, 4), dtype=torch.int16, device=dev)

    # Data plane of the row-tiled frame: the C-ABI's RCCL collectives. Should the communicator not come up on some rank
    # (all ranks agree through the control group), torch.distributed's own RCCL backend takes over: same collectives.
    comm, torch_group, collective_api = None, None, None
    if (world > 1 or args.force_tiled) and args.backend == "nccl":
        ok = 1
        try:
            comm = rowtile.Comm(rank, world, local_rank)
            collective_api = "szg_rowtile_comm (C-ABI, RCCL)"
        except Exception as e:
            if getattr(e, "code", None) == abi.SZG_ERR_TIMEOUT:
                # the rendezvous missed its deadline (SZG_COMM_TIMEOUT_S): ncclCommInitRank is still blocked on a helper thread
                # and cannot be cancelled. Leave with a non-zero status NOW - that is what makes the launcher tear the other
                # ranks down - without destructors, without another collective, without re-executing anything.
                log(f"rank {rank}: {e}; exiting with status 3")
                sys.stderr.flush()
                os._exit(3)
            log(f"rank {rank}: szg_rowtile_comm failed ({e}); falling back to torch.distributed's RCCL backend")
            ok = 0
        agreed = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        if int(agreed.item()) == 0:
            if comm is not None:
                comm.destroy()
                comm = None
            torch_group = dist.new_group(backend="nccl")
            collective_api = "torch.distributed (RCCL)"
    elif world > 1 or args.force_tiled:
        collective_api = "torch.distributed (gloo, host-staged rehearsal)"
    sky_lut = sky.skyviewLUT_tensor() if (tiled and comm is None) else None
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(args.steps)]
    spot_arg = spots if SPOTS else None
    pending = []  # (gathered buffer, work) of the previous frame's tile gather

    # rank 0 scatters the gathered tiles into the frame on a side stream: an HBM-bound copy that runs under the next
    # frame's VALU-bound passes instead of in front of them (rank 0 is the slowest rank otherwise)
    side = torch.cuda.Stream() if (tiled and rank == 0) else None
    compose_done = {}

    def finish_gather():
        while pending:
            buf, work = pending.pop(0)
            work.wait()  # the compute stream waits for the collective; the host does not block
            if rank == 0:
                arrived = torch.cuda.Event()
                arrived.record()
                side.wait_event(arrived)
                rowtile.compose(buf, H, nranks, BLOCK_ROWS, out=composed, stream=side.cuda_stream)
                done = torch.cuda.Event()
                done.record(side)
                compose_done[buf.data_ptr()] = done

    def frame(k, events=None):
        e = events
        tgt = targets[k % len(targets)]
        if e:
            e[0].record()
        lut_work = None
        if tiled:
            # LUTs first: every rank computes 1/N of the sky-view LUT rows, and the all-gather of the slices runs
            # while the lights pass (which needs no LUT) is shading
            sky.recordTransmittance(None, 0, atmospheres)
            if sky.desc.skyview_height % nranks != 0:
                # the LUT's rows do not divide over the ranks: every rank computes the whole LUT (no second collective)
                sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
                lut_work = rowtile._Done()
            else:
                b, en = sky.lutRowSlice(rank, nranks)
                sky.recordSkyViewLUTRows(None, 0, atmospheres, 0, cameras, b, en)
                if comm is not None:
                    lut_work = comm.allgather_skyview_lut(sky)
                else:
                    lut_work = rowtile.allgather_lut(sky_lut, rank, nranks, group=torch_group, async_op=True, force=True)
                    sky.invalidateLUTs(abi.SZG_LUT_SKYVIEW)  # texels written through the aliased tensor
            if e:
                e[1].record()
                e[2].record()
            deferred.recordLights(None, rect, tgt, 1, lights, spot_arg, 0, cameras, tile=tile)
            if e:
                e[3].record()
            lut_work.wait()
        else:
            deferred.recordLights(None, rect, tgt, 1, lights, spot_arg, 0, cameras, tile=tile)
            if e:
                e[1].record()
            sky.recordTransmittance(None, 0, atmospheres)
            if e:
                e[2].record()
            sky.recordSkyViewLUT(None, 0, atmospheres, 0, cameras)
            if e:
                e[3].record()
        sky.recordComposite(None, tgt, rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0,
                            lights, tile=tile)
        if e:
            e[4].record()
        if tiled:
            # the gather of frame k-1 has had a whole frame of compute to finish: compose it now, then start ours
            finish_gather()
            if rank == 0 and gathered[k % 2].data_ptr() in compose_done:
                # this gather buffer was last read by the compose of frame k-2 on the side stream
                torch.cuda.current_stream().wait_event(compose_done[gathered[k % 2].data_ptr()])
            if comm is not None:
                buf, work = comm.gather_tiles(tgt.color, gathered=gathered[k % 2] if rank == 0 else None, dst=0)
            else:
                buf, work = rowtile.gather_tiles(tgt.color, rank, nranks, gathered=gathered[k % 2] if rank == 0 else None,
                                                 group=torch_group, async_op=True)
            pending.append((buf, work))
        if e:
            e[5].record()

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for k in range(args.warmup):
        frame(k)
    finish_gather()
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        frame(k, ev[k])
    finish_gather()  # the last frame's gather + compose belong to the timed region
    sync_all()
    elapsed = time.perf_counter() - t0

    t = torch.tensor([elapsed], dtype=torch.float64)
    g = torch.tensor([geometry_px_local], dtype=torch.int64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        if tiled:
            dist.all_reduce(g, op=dist.ReduceOp.SUM)
    elapsed = float(t.item())
    frames_total = args.steps * (1 if (tiled or world == 1) else world)  # replicas render `world` frames per step
    total_px = W * H * frames_total
    value = total_px / elapsed / 1e6

    # per-pass device times on this rank (ms), from the events recorded inside the timed region
    # (tiled runs record the events in the order luts, -, lights, composite: see frame())
    names = ["lights", "transmittance", "skyview", "composite", "gather+compose"]
    if tiled:
        names = ["transmittance+skyview_rows", "-", "lights(+lut allgather)", "composite", "compose(k-1)+gather start"]
    per = {n: float(np.mean([ev[k][i].elapsed_time(ev[k][i + 1]) for k in range(args.steps)])) for i, n in enumerate(names)}
    G_frame = int(g.item()) if tiled else geometry_px_local
    N_local = W * rows
    G_local = geometry_px_local
    # algorithmic bytes: composite reads G-buffer 48 + depth 4 per px, prior colour 8 per geometry px,
    # writes 8 per px; lights reads 48 per px and writes 8 per px (clear fused). Per launch on this rank.
    bytes_composite = 60 * N_local + 8 * G_local
    bytes_lights = 56 * N_local
    bytes_luts = 16 * (512 * 128 + 2048 * 1024)
    # `roofline` describes the DOMINANT kernel of THIS run: the longest pass by its mean launch time, measured with events
    # on the launch stream inside the timed region. Algorithmic bytes per launch: SURVEY 8d / DESIGN.md 4.
    sky_texels = sky.desc.skyview_width * sky.desc.skyview_height
    candidates = {
        "k_composite": (per["composite"], bytes_composite, "60 B/px + 8 B per geometry px"),
        "k_lights": (per[names[2] if tiled else "lights"], bytes_lights, "56 B/px (48 G-buffer read + 8 colour written, clear fused)"),
    }
    if not tiled:
        candidates["k_skyview"] = (per["skyview"], 16 * sky_texels, "16 B per LUT texel written")
    dominant = max(candidates, key=lambda k: candidates[k][0])
    dom_ms, dom_bytes, dom_rule = candidates[dominant]
    dom_s = dom_ms / 1e3
    roofline = {
        "bound": "hbm", "kernel": dominant,
        "achieved": dom_bytes / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": dom_bytes / dom_s / 1e9 / HBM_PEAK_GBS,
        "traffic": None,
        "algorithmic_bytes_per_launch": dom_bytes, "algorithmic_bytes_rule": dom_rule, "avg_launch_ms": dom_ms,
    }
    # HBM traffic and the VALU picture come from counters, which cannot be read inside this run: they are taken from the
    # committed profile of THIS workload (tools/collect_pmc.py: rocprofv3, separate --pmc passes, gfx950 FETCH_SIZE
    # correction) and only when that file names this workload and this kernel.
    roofline_valu = None
    pmc_path = latest_profile(f"pmc_{name}.json")
    pmc_rel = os.path.relpath(pmc_path, ROOT) if pmc_path else None
    if pmc_path and not tiled:
        try:
            prof = json.load(open(pmc_path))
            k = prof["kernels"].get(dominant) if prof.get("workload") == name else None
            if k is not None:
                roofline["traffic"] = k.get("hbm_bytes_per_launch")
                roofline["traffic_source"] = pmc_rel
                # which tree the counter profile was collected from, and what the kernel took there: counters cannot be
                # read inside this run, so the line says how far the profile is from what it decorates
                roofline["traffic_profile"] = {"source_hash": prof.get("source_hash"), "avg_launch_ms": k.get("avg_launch_ms"),
                                               "same_sources_as_this_run": prof.get("source_hash") == entry.source_hash("hip")}
                if "valu_issue_cycles_per_launch" in k:
                    # every VALU instruction class priced at its issue cost measured by tools/opcost.hip
                    # (profiles/r02_opcost.txt): what the kernel needs of the 1024 SIMDs if it did nothing but issue
                    cyc = k["valu_issue_cycles_per_launch"]
                    simds = 1024
                    roofline_valu = {
                        "bound": "valu-issue", "kernel": dominant, "unit": "SIMD-cycles per launch",
                        "instructions_per_launch": k.get("SQ_INSTS_VALU"), "issue_cycles_per_launch": cyc,
                        "class_costs": prof.get("class_issue_cycles"),
                        "clock_GHz_at_which_issue_alone_takes_the_measured_time": cyc / simds / dom_s / 1e9,
                        "frac_of_issue_ceiling_at_2.4GHz": cyc / simds / 2.4e9 / dom_s,
                        "source": f"{pmc_rel} + profiles/r02_opcost.txt",
                    }
        except Exception as e:  # a malformed profile must not cost the bench line
            log("profile not usable:", e)
    frame_bytes = bytes_composite + bytes_lights + bytes_luts
    frame_s = sum(per[n] for n in names[:4]) / 1e3
    if tiled:
        per.setdefault("lights", per[names[2]])
        per.setdefault("transmittance", 0.0)
        per.setdefault("skyview", per[names[0]])

    out = {
        "metric": "Mpixels/s deferred-lit+atmosphere", "value": value, "unit": "Mpixels/s", "n_gpus": args.gpus,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
        "scaling": "strong" if wl["tiled"] else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": wl["label"], "width": W, "height": H, "spot_lights": SPOTS, "geometry_fraction": G_frame / (W * H),
                   "sun_elevation_deg": args.elevation, "row_tile_block_rows": BLOCK_ROWS if tiled else None,
                   "parallelism": (f"rowtile{nranks}+gather" if tiled else ("single" if world == 1 else f"replicas{world}")),
                   "collectives": (args.backend if (tiled and (world > 1 or args.force_tiled)) else None),  # data-path collectives only
                   "collective_api": collective_api, "rccl_ranks": (comm.size() if comm is not None else None),
                   "rccl_bound": (lib().szg_rowtile_comm_backend().decode() if comm is not None else None)},
        "pass_ms_rank0": per,
        "source_hash": entry.source_hash("hip"),  # the sources on disk ...
        "library_build_id": lib().szg_build_id().decode(),  # ... and what the loaded binary says it was built from
        "roofline": roofline,
        "roofline_valu": roofline_valu,
        "roofline_frame": {"algorithmic_bytes": frame_bytes, "device_ms": frame_s * 1e3,
                           "achieved_GBps": frame_bytes / frame_s / 1e9, "frac": frame_bytes / frame_s / 1e9 / HBM_PEAK_GBS},
    }

    if tiled:
        # N = 1 of the default run is the 4K workload the metric is quoted on; the strong-scaling base of THIS workload
        # (same 8K frame on one GPU, untiled) is a committed measurement, repeated here so that the speed-up can be read
        # off the line without mixing workloads.
        ref = latest_profile("bench_c4.json")
        if ref:
            try:
                r1 = json.load(open(ref))
                out["strong_scaling_reference"] = {"workload": name, "n_gpus": 1, "value": r1["value"], "ms_per_step": r1["ms_per_step"],
                                                   "image_checksum": r1.get("image_checksum"), "source": os.path.relpath(ref, ROOT),
                                                   "speedup_vs_reference": value / r1["value"]}
            except Exception:
                pass

    # Checksum of the final RGBA16 image (rank 0: the composed frame when tiled), outside the timed region: equal for
    # every N and for the untiled run of the same workload — the row-tiled path changes no pixel.
    def checksum(final):
        img = final.view(torch.int16).to(torch.int64) & 0xFFFF
        weights = (torch.arange(H, device=img.device, dtype=torch.int64) % 251 + 1).view(H, 1, 1)
        return {"sum": int(img.sum().item()), "row_weighted_sum": int((img * weights).sum().item())}

    if rank == 0:
        out["image_checksum"] = checksum(composed if tiled else targets[(args.steps - 1) % len(targets)].color[:H])

    extras = world == 1 and not tiled and not args.no_extras
    if extras:
        # LUT reuse across frames (abi.h szg_skyview_set_lut_reuse; SURVEY 8e: identical results): the steady state of a
        # scene whose atmosphere, sun and camera do not move. NOT part of `value` - the reference recomputes both LUTs every
        # frame and so does the timed region above.
        sky.setLUTReuse(True)
        for k in range(2):
            frame(k)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for k in range(args.steps):
            frame(k)
        torch.cuda.synchronize()
        cached = (time.perf_counter() - c0) / args.steps
        sky.setLUTReuse(False)
        out["steady_state_cached_luts"] = {"in_value": False, "ms_per_frame": cached * 1e3, "mpixels_per_s": W * H / cached / 1e6,
                                           "same_image": checksum(targets[(args.steps - 1) % len(targets)].color[:H]) == out["image_checksum"],
                                           "note": "both LUT passes skipped on the device while their parameter blocks are bit-equal to "
                                                   "those the texels were computed from; same image"}

    if extras:
        # The pass right after the path, measured on its own outside the timed region: in-place
        # OETF of the final image, 16 B/px. Not part of `value`.
        # Rotate over enough distinct images (> 256 MiB together) that no pass finds its image in the Infinity Cache.
        n_img = max(2, int(300e6 // (8 * W * H)) + 1)
        images = [pl.SceneTexture(W, H, dev) for _ in range(n_img)]
        for im in images:
            im.color.copy_(target.color[:H])
        reps = 4 * n_img
        for im in images:
            pl.recordOETF(None, im, W, H)
        torch.cuda.synchronize()
        o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o0.record()
        for r in range(reps):
            pl.recordOETF(None, images[r % n_img], W, H)
        o1.record()
        torch.cuda.synchronize()
        oetf_ms = o0.elapsed_time(o1) / reps
        del images
        out["oetf_pass"] = {"kernel": "k_oetf", "ms": oetf_ms, "algorithmic_bytes": 16 * W * H,
                            "achieved_GBps": 16 * W * H / (oetf_ms / 1e3) / 1e9,
                            "frac_of_8TBps": 16 * W * H / (oetf_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "in_value": False,
                            "note": "65536-entry transfer table; images rotated so that every pass streams from HBM"}

    if extras:
        # Extension without a reference counterpart (abi.h "Aerial-perspective froxel LUT"): APPROXIMATE composite that
        # replaces the inline per-pixel march by a froxel-LUT fetch. Reported separately, never part of `value`.
        reps = 10
        f0, f1, f2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        sky.recordAerialLUT(None, 0, atmospheres, 0, cameras, 0.032)
        torch.cuda.synchronize()
        f0.record()
        for _ in range(reps):
            sky.recordAerialLUT(None, 0, atmospheres, 0, cameras, 0.032)
        f1.record()
        for _ in range(reps):
            sky.recordCompositeFast(None, target, rect, deferred.gbuffer(), deferred.shadowMaps(), 0, atmospheres, 0, cameras, 0, lights)
        f2.record()
        torch.cuda.synchronize()
        aerial_ms, fast_ms = f0.elapsed_time(f1) / reps, f1.elapsed_time(f2) / reps
        # BASELINE.json names "all 4 LUTs": the reference has neither a multi-scatter nor an aerial-perspective LUT pass
        # (SURVEY a17 / a18), so the exact frame computes two LUTs and `value` times exactly the reference's four passes.
        # What the frame costs when the other two LUTs are computed as well (no pixel changes: the exact composite does
        # not read them) is stated here.
        sky.recordMultiScatterLUT(None, 0, atmospheres)
        torch.cuda.synchronize()
        m0, m1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        m0.record()
        for _ in range(reps):
            sky.recordMultiScatterLUT(None, 0, atmospheres)
        m1.record()
        torch.cuda.synchronize()
        multi_ms = m0.elapsed_time(m1) / reps
        frame4 = elapsed / args.steps * 1e3 + multi_ms + aerial_ms
        out["all_four_luts"] = {"in_value": False, "multiscatter_lut_ms": multi_ms, "aerial_lut_ms": aerial_ms,
                                "frame_ms_with_all_four_luts": frame4, "mpixels_per_s_with_all_four_luts": W * H / frame4 / 1e3,
                                "note": "exact frame + the two LUT passes the reference does not have (opt-in extensions)"}
        out["fast_composite_extension"] = {
            "approximate": True, "in_value": False, "aerial_lut_ms": aerial_ms, "composite_fast_ms": fast_ms,
            "achieved_GBps": bytes_composite / (fast_ms / 1e3) / 1e9, "frac_of_8TBps": bytes_composite / (fast_ms / 1e3) / 1e9 / HBM_PEAK_GBS,
            "frame_ms_with_fast_mode": per["lights"] + per["transmittance"] + per["skyview"] + aerial_ms + fast_ms}

    if extras:
        # The pass right before the path: the G-buffer raster of the same scene given as real meshes
        # (instanced cubes + ground quad) into a second target. Producer of the path's inputs: never part of `value`.
        from syzygy_amd import meshes as mesh_lib

        scene_meshes = mesh_lib.meshes_of_fill_scene(syn.fill)
        raster_target = pl.SceneTexture(W, H, dev)
        raster_pipe = pl.DeferredShadingPipeline((W, H), max_spot_lights=1, max_shadow_maps=0, device_index=local_rank)
        marr = mesh_lib.mesh_array(scene_meshes, dev)
        st = raster_target.abi()
        from syzygy_amd import lib as _lib
        from syzygy_amd._lib import check as _check

        def raster_once():
            _check(_lib().szg_deferred_record_gbuffer_raster(raster_pipe._h, pl._stream_handle(None), rect, None, C.byref(st), 0,
                                                             C.c_void_p(cameras.deviceAddress()), marr, len(scene_meshes)))

        raster_once()
        torch.cuda.synchronize()
        reps = 10
        r0, r1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        r0.record()
        for _ in range(reps):
            raster_once()
        r1.record()
        torch.cuda.synchronize()
        raster_ms = r0.elapsed_time(r1) / reps
        prims = sum(len(m.indices) // 3 * len(m.models) for m in scene_meshes)
        out["gbuffer_raster_pass"] = {"kernels": "k_raster_setup + k_raster_chunks + k_raster_superchunks + k_raster_tile", "primitives": prims,
                                      "ms": raster_ms, "algorithmic_bytes": 52 * W * H,
                                      "achieved_GBps": 52 * W * H / (raster_ms / 1e3) / 1e9,
                                      "frac_of_8TBps": 52 * W * H / (raster_ms / 1e3) / 1e9 / HBM_PEAK_GBS, "in_value": False}
        raster_pipe.cleanup()

    if rank == 0 and extras:
        # How far the library that was TIMED above is from the reference: tests/gpu_spirv_pin_child.py renders the inputs of
        # tests/golden/spirv_vectors.npz (5 248 values the reference's committed SPIR-V computes when executed literally) with
        # the same library, in a child process. north_star's bar is 1e-4 relative; the product's contraction rule
        # (include/szg/contraction.h) is chosen so that it holds, with one UNORM16 step on the stored image.
        import subprocess

        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "gpu_spirv_pin_child.py")], env=dict(os.environ),
                               capture_output=True, text=True, timeout=300)
            pin = json.loads(r.stdout.strip().split("\n")[-1])
            out["parity_vs_spirv_vectors"] = {
                "library": pin["library"], "values": pin["values"],
                "rel_max": max(pin[k + "_rel_max"] for k in ("transmittance", "skyview", "lights", "camera")),
                "lsb_max": max(pin["lights_unorm_max_step"], pin["camera_unorm_max_step"]),
                "per_shader_rel_max": {k: pin[k + "_rel_max"] for k in ("transmittance", "skyview", "lights", "camera")},
                "bit_identical_values": pin["values"] - sum(pin[k + "_mismatches"] for k in ("transmittance", "skyview", "lights", "camera")),
                "rel": "|got - want| / max(|want|, 1e-3)", "vectors": "tests/golden/spirv_vectors.npz",
                "within_1e-4_and_1_lsb": bool(max(pin[k + "_rel_max"] for k in ("transmittance", "skyview", "lights", "camera")) <= 1e-4
                                              and max(pin["lights_unorm_max_step"], pin["camera_unorm_max_step"]) <= 1)}
        except Exception as e:  # the extra must never take the bench line down
            log(f"parity_vs_spirv_vectors extra failed: {e}")

    if extras and "SZG_HIP_LIBRARY" not in os.environ:
        # The literal build of the same kernels (libszg_hip_literal.so: the contraction rule switched off, bit-identical to
        # the reference's SPIR-V executed literally, tests/test_gpu_spirv_pin.py), timed by a child process on the same
        # workload. NOT part of `value`: it says what the one switch costs and how far the two images are apart.
        import subprocess

        literal = os.path.join(ROOT, "syzygy_amd", "csrc", "libszg_hip_literal.so")
        if os.path.exists(literal):
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--workload", name, "--steps", str(args.steps), "--warmup",
                                    str(args.warmup), "--no-cpu-baseline", "--no-extras"], env=dict(os.environ, SZG_HIP_LIBRARY=literal),
                                   capture_output=True, text=True, timeout=300)
                lit = json.loads(r.stdout.strip().split("\n")[-1])
                out["literal_build"] = {"in_value": False, "library": "libszg_hip_literal.so", "ms_per_frame": lit["ms_per_step"],
                                        "mpixels_per_s": lit["value"], "image_checksum": lit["image_checksum"],
                                        "note": "same kernels with two roundings at the places where the product fuses a * b + c; "
                                                "reproduces the reference's SPIR-V vectors bit for bit (tests/test_gpu_spirv_pin.py)"}
            except Exception as e:  # the extra must never take the bench line down
                log(f"literal_build extra failed: {e}")

    if rank == 0 and not args.no_cpu_baseline and args.gpus == 1:
        out["cpu_baseline"] = cpu_baseline(args, wl, atm, cam, sun, moon, spots, syn)
    if rank == 0:
        log("per-pass device ms:", {k: round(v, 4) for k, v in per.items()})
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    deferred.cleanup()
    sky.destroy()


def latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round that has one (None if no round has)."""
    import glob

    found = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{suffix}")))
    return found[-1] if found else None


def self_launch(nproc, real_stdout):
    """Start `nproc` ranks of this script under torch.distributed.run (one process per GPU, rendezvous on 127.0.0.1 at a
    free port), pass the one JSON line of rank 0 through to the saved stdout descriptor and return the launcher's status."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch:", " ".join(cmd))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, env=env)
    lines = 0
    for raw in child.stdout:
        text = raw.decode("utf-8", "replace")
        if text.lstrip().startswith('{"metric"'):
            os.write(real_stdout, raw)
            lines += 1
        else:
            sys.stderr.write(text)
    status = child.wait()
    if status == 0 and lines != 1:
        log(f"self-launch: expected one JSON line from rank 0, saw {lines}")
        return 1
    return status


def cpu_baseline(args, wl, atm, cam, sun, moon, spots, syn):
    """The scalar oracle (oracle/, the parity checker) timed on this box's cores on a bounded
    sample of the same workload: both LUTs in full, plus lights + composite on `cpu_rows` rows
    spread evenly over the frame (row-tile with 1-row blocks), scaled to the full height."""
    from oracle import binding as ob
    from syzygy_amd import abi, lib

    W, H, SPOTS = wl["width"], wl["height"], wl["spots"]
    threads = max(1, min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 64))
    sample_rows = min(args.cpu_rows, H)
    nranks = H // sample_rows
    tile = abi.RowTile(1, nranks // 2, nranks, lib().szg_rowtile_local_rows(H, 1, nranks // 2, nranks))
    rect = abi.Rect(0, 0, W, H)
    frame = ob.HostFrame(W, tile.local_rows, debug=False)
    dirs = (abi.DirectionalLightPacked * 2)(sun, moon)
    ob.gbuffer_fill(frame, rect, tile, cam, syn.fill, threads=threads)
    t0 = time.perf_counter()
    tlut = ob.transmittance_lut(atm, 512, 128, threads=threads)
    t1 = time.perf_counter()
    slut = ob.skyview_lut(atm, cam, tlut, 2048, 1024, threads=threads)
    t2 = time.perf_counter()
    ob.lights(frame, rect, tile, None, cam, dirs, 2, 1, spots, SPOTS, threads=threads)
    t3 = time.perf_counter()
    ob.composite(frame, rect, tile, None, atm, cam, dirs, 0, tlut, slut, threads=threads)
    t4 = time.perf_counter()
    scale = H / tile.local_rows
    est = (t1 - t0) + (t2 - t1) + ((t3 - t2) + (t4 - t3)) * scale

    # (i) of BASELINE.md: the same restatement on ONE thread, on a smaller sample of the same frame: a quarter of the
    # transmittance LUT's texels (a 256 x 64 LUT: config 1's own size), 8 sky-view LUT rows above and 8 below the horizon,
    # lights + composite on 24 rows spread evenly over the frame.
    s0 = time.perf_counter()
    ob.transmittance_lut(atm, 256, 64, threads=1)
    s1 = time.perf_counter()
    scratch = np.zeros((1024, 2048, 4), np.float32)
    ob.skyview_lut(atm, cam, tlut, 2048, 1024, row_begin=252, row_end=260, threads=1, out=scratch)
    ob.skyview_lut(atm, cam, tlut, 2048, 1024, row_begin=764, row_end=772, threads=1, out=scratch)
    s2 = time.perf_counter()
    n1 = H // 24
    tile1 = abi.RowTile(1, n1 // 2, n1, lib().szg_rowtile_local_rows(H, 1, n1 // 2, n1))
    frame1 = ob.HostFrame(W, tile1.local_rows, debug=False)
    ob.gbuffer_fill(frame1, rect, tile1, cam, syn.fill, threads=threads)
    s3 = time.perf_counter()
    ob.lights(frame1, rect, tile1, None, cam, dirs, 2, 1, spots, SPOTS, threads=1)
    s4 = time.perf_counter()
    ob.composite(frame1, rect, tile1, None, atm, cam, dirs, 0, tlut, slut, threads=1)
    s5 = time.perf_counter()
    est1 = (s1 - s0) * 4.0 + (s2 - s1) * (1024 / 16) + ((s4 - s3) + (s5 - s4)) * (H / tile1.local_rows)
    return {
        "value": W * H / est / 1e6, "unit": "Mpixels/s", "cores": threads, "kind": "port",
        "sample": (f"both LUTs in full ({t1 - t0:.2f}s + {t2 - t1:.2f}s) + lights/composite on {tile.local_rows} of {H} rows "
                   f"spread evenly over the frame ({t3 - t2:.2f}s + {t4 - t3:.2f}s), extrapolated x{scale:.1f}; "
                   f"{threads} std::threads, -O2 -ffp-contract=off"),
        "estimated_frame_s": est,
        "single_thread": {
            "value": W * H / est1 / 1e6, "unit": "Mpixels/s", "cores": 1, "kind": "port", "estimated_frame_s": est1,
            "sample": (f"transmittance LUT 256x64 ({s1 - s0:.2f}s, x4 texels), 16 of 1024 sky-view LUT rows ({s2 - s1:.2f}s), "
                       f"lights/composite on {tile1.local_rows} of {H} rows ({s4 - s3:.2f}s + {s5 - s4:.2f}s); one thread"),
        },
    }


if __name__ == "__main__":
    main()
