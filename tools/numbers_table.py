#!/usr/bin/env python
"""Prints the DESIGN.md section-7 table from the committed round files (profiles/<round>_bench_*.json, <round>_*_pmc.json)."""
import json, os, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
def J(name):
    return json.load(open(os.path.join(P, f"{R}_{name}.json")))
def pmc(wl, B, units):
    j = J(f"{wl}_b{B}_pmc")["pmc_mean_per_launch"]
    out = []
    for k, v in j.items():
        if "FETCH_SIZE" not in v: continue
        by = (2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024; us = list(v["_launch_us"].values())[0]
        out.append((k.split("<")[0].replace("void ", "").replace("mw::", ""), us, by / units, by / us / 1e6, 2 * v["FETCH_SIZE"] * 1024 / units, v["WRITE_SIZE"] * 1024 / units))
    return out
def row(title, b, extra=""):
    r = b["roofline"]
    ws = b.get("hbm_roofline_frac_whole_step")
    print(f"| {title} | {b['value']:.3g} {b['unit'].split('/')[0]}/s, {b['ms_per_step'] * 1e3:.2f} µs/step"
          + (f" (median of {b['repeats']}; {b['ms_per_step_min'] * 1e3:.2f} … {b['ms_per_step_max'] * 1e3:.2f}; HIP events {b['event_ms_per_step'] * 1e3:.2f})" if b.get("event_ms_per_step") else "")
          + f" | {'%.3f' % ws if ws else ''} | {extra} |")
print(f"build {J('bench_ocean1024_driver_k20')['build_id']}")
for name, wl, B, NN in (("bench_ocean1024_driver_k20", "ocean1024", 20, 1024 ** 2), ("bench_ocean1024_b32_steps640", "ocean1024", 32, 1024 ** 2),
                        ("bench_ocean1024_tiles_gather", None, 32, 1024 ** 2), ("bench_ocean2048", "ocean2048", 32, 2048 ** 2), ("bench_ocean4096", "ocean4096", 32, 4096 ** 2)):
    b = J(name); r = b["roofline"]
    ex = f"`k_pass2_hs` {r['launch_us']:.1f} µs = **{r['frac']:.3f}**; `k_pass1` {r['kernels'][0]['us_per_launch']:.1f} µs"
    if wl:
        for k, us, bpp, tbs, rd, wr in pmc(wl, B, NN * B):
            ex += f"; PMC {k} {us:.1f} µs, {bpp:.1f} B/pt ({rd:.1f} read + {wr:.1f} written) at {tbs:.2f} TB/s"
    if b.get("with_gather"): ex += f"; with the gather {b['with_gather']['value']:.3g}"
    if b.get("single_step_us"): ex += f"; one step per call {b['single_step_us']:.1f} µs, host arrays {b['frame_at_a_time']['host_ms_per_frame_registered']:.2f} ms"
    row(name, b, ex)
b = J("bench_pond"); row("pond", b, f"frac {b['roofline']['frac']:.3f}; " + "; ".join(f"PMC {k} {us:.1f} µs {bpp:.2f} B/vertex-step {tbs:.2f} TB/s" for k, us, bpp, tbs, rd, wr in pmc("pond", 32, 32e6)))
for name, B, frames_per_launch in (("bench_renderer1024", 32, None), ("bench_renderer1024_frame", 1, 1), ("bench_renderer1024_tiles4", 4, 4)):
    try:
        b = J(name)
    except Exception as ex:
        print(f"| {name} | missing ({ex}) | | |"); continue
    wf = b["roofline"].get("whole_frame") or {}
    ex = (f"{b['roofline']['kernel'][:28]} frac {b['roofline']['frac']:.3f}; whole frame {wf.get('frac', float('nan')):.3f} of {wf.get('algorithmic_bytes_per_texel_frame', 0):.1f} B "
          f"(plan {wf.get('plan_bytes_per_texel_frame', 0):.1f} B; physical {b['roofline'].get('physical_bytes_per_texel_frame') or float('nan'):.1f} B); ")
    ex += "; ".join(f"{k['name'][:18]} {k['us_per_launch']:.1f} µs" for k in (b["roofline"].get("kernels") or []))
    try:    # per launch: a steps call launches pass 2 / the normal pass per chunk of 8 frames, the spectrum kernel once per 32
        ex += " | PMC per launch: " + "; ".join(f"{k} {us:.1f} µs {by / 1e6:.1f} MB {tbs:.1f} TB/s" for k, us, by, tbs, rd, wr in pmc("renderer1024", B, 1.0) if "pass" in k or "normal" in k)
    except Exception:
        pass
    row(name, b, ex)
for n in (12, 50, 100, 1000, 2000):
    b = J(f"bench_direct_{n}"); print(f"| direct N={n} | {b['value']:.3g} pts/s {b['ms_per_step'] * 1e3:.1f} µs/step | | {[(k['name'][:12], round(k['us_per_step'], 1)) for k in b['roofline']['kernels']]} |")
c = J("bench_ocean1024_driver_k20")["cpu_baseline"]
print("cpu:", c["value"], c["sample"][:60], {k: (v.get("value") if isinstance(v, dict) else v) for k, v in c.items() if k in ("fft_port", "fft_port_c", "fft_port_c_threads")})
