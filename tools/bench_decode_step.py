#!/usr/bin/env python3
"""Tuning aid: lock-step decode steps at a FIXED context length, on one or several concurrent lanes (cz_debug_decode_steps).
Each config is `name:B=<streams per lane>,L=<lanes>,pos=<context>,x=<0|1: one XCD per lane via a CU-masked stream>,av=<dbg_skip: 1 = no attention, 2 = attention only>,
fn=<fuse_norm>,gt=<dec_grid_target>,dw=<dec_wide_rows: dec_gemm blocks of 32 x 64 from this many rows on>,wm=<wide_min_streams: groups at least this wide take the tiled-GEMM step>,nw=<mid_nw: K groups of gemm_mid, 0 = by K>,mq=<mid_quad>,sp=<dec_attn_split_max>,gv=<gemv_max_rows: groups of at most this many streams take the VALU projections, 0 = never>,pad=<dec_attn_lds_pad: extra LDS bytes per decode-attention block>,mt=<dec_mt_max>,g=<1 graph|0 eager>,prof=<1: eager per-family HIP events on lane 0>`.  Prints one JSON line per config."""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import candlezip_amd as cz


WM = [256, 1]
STEP_SYNC = [0]


def run_config(parent, cfgs, name, B, L, pos, x, av, fn, g, prof, iters, sym, av1=-1, pr0=0, pr1=0):
    lib = cz.lib()
    lib.cz_debug_decode_steps.restype = C.c_int
    lib.cz_debug_decode_steps.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
    lanes = [parent.fork() for _ in range(L)]
    for k, s in enumerate(lanes):
        s.debug_set("max_lanes", 1)
        if x:
            nx = max(1, 8 // L)
            s.debug_set("cu_mask_xcd", (k * nx) % 8 + 16 * (nx - 1))
        s.debug_set("dbg_skip", av1 if (k & 1) and av1 >= 0 else av)       # av1: the odd lanes' dbg_skip (attention-only lanes beside chain-only lanes)
        if (pr1 if k & 1 else pr0):
            s.debug_set("stream_priority", pr1 if k & 1 else pr0)           # -1 = high, 1 = low
        s.debug_set("fuse_norm", fn)
        s.debug_set("step_sync", STEP_SYNC[0])
        s.debug_set("wide_min_streams", WM[0]); s.debug_set("wide_mode", WM[1])
    ms = [0.0] * L; err = []
    gate = threading.Barrier(L)

    def worker(k):
        out = C.c_double()
        gk = 0 if (prof and k == 0) else g
        rc = lib.cz_debug_decode_steps(lanes[k].h, B, pos, 0, gk, sym, C.byref(out))       # prepare: buffers, graph capture, warm step
        if rc:
            err.append(lib.cz_last_error().decode())
        if prof and k == 0:
            lanes[k].profile_enable(True)
        gate.wait()                                                                          # every lane starts its timed loop together
        if not rc:
            rc = lib.cz_debug_decode_steps(lanes[k].h, B, pos, iters, gk, sym, C.byref(out))
            if rc:
                err.append(lib.cz_last_error().decode())
        ms[k] = out.value

    th = [threading.Thread(target=worker, args=(k,)) for k in range(L)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    res = {"config": name, "B": B, "lanes": L, "pos": pos, "xcd_lanes": x, "dbg_skip": av, "fuse_norm": fn, "graph": g, "iters": iters,
           "ms_per_step_lane": [round(m / iters, 4) for m in ms], "tokens_per_s_device_time": round(L * B * iters / (max(ms) * 1e-3), 1) if not err else None,
           "wall_s_incl_setup": round(wall, 2), "errors": err}
    if prof:
        res["families_us_per_step_lane0"] = {k: round(1e3 * v["total_ms"] / iters, 2) for k, v in lanes[0].profile_read().items()}
    for s in lanes:
        s.close()
    print(json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--max-streams", type=int, default=128)
    ap.add_argument("--sym", type=int, default=1, help="include the decode_symbol (CDF) kernel in the step")
    ap.add_argument("configs", nargs="+")
    args = ap.parse_args()
    parent = cz.Session("smollm", cz.smollm2_135m_config(), seed=0x5EED, embed_std=0.02, max_streams=args.max_streams)
    for c in args.configs:
        name, _, kv = c.partition(":")
        d = dict(B=128, L=1, pos=767, x=0, av=0, fn=0, g=1, prof=0, gt=384, mt=8, wm=256, dw=256, wmode=1, nw=4, pad=0, mq=1, sp=36, av1=-1, pr0=0, pr1=0, gv=4, fa=1, ss=0, aw=8)
        for item in kv.split(","):
            if item:
                k, v = item.split("=")
                d[k] = int(v)
        parent.debug_set("dec_grid_target", d["gt"]); parent.debug_set("dec_mt_max", d["mt"])     # (process-wide knobs)
        STEP_SYNC[0] = d["ss"]                                                # ss=1: one host wait per step (eager runs under rocprofv3)
        WM[0] = d["wm"]; WM[1] = d["wmode"]; parent.debug_set("dec_wide_rows", d["dw"]); parent.debug_set("mid_nw", d["nw"]); parent.debug_set("dec_attn_lds_pad", d["pad"])
        parent.debug_set("gemv_max_rows", d["gv"]); parent.debug_set("gemv_fold_attn", d["fa"])      # fa: the o-projection adds up the split attention's partials
        parent.debug_set("dec_attn_waves", d["aw"])                           # aw: waves per block of the whole-device decode attention (8, or 4: four blocks per CU)
        parent.debug_set("mid_quad", d["mq"]); parent.debug_set("dec_attn_split_max", d["sp"])     # mq: gemm_mid blocks share sum + epilogue over their waves; sp: split attention up to this many (stream, head) pairs
        run_config(parent, None, name + f"[gt={d['gt']}]", d["B"], d["L"], d["pos"], d["x"], d["av"], d["fn"], d["g"], d["prof"], args.iters, args.sym, d["av1"], d["pr0"], d["pr1"])
    parent.close()


if __name__ == "__main__":
    main()
