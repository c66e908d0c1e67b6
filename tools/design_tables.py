#!/usr/bin/env python3
"""Write the measurements table of DESIGN.md from the committed profiles (round 4's verdict: numbers live in profiles/ and are CITED,
not copied by hand - DESIGN.md, README.md and raptor_quad.h disagreed in the last digit).

    python tools/design_tables.py r05          rewrites the block between the GENERATED markers in DESIGN.md

Every row names the file and the field it was read from; nothing else in DESIGN.md is a measurement."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- BEGIN GENERATED MEASUREMENTS (tools/design_tables.py) -->", "<!-- END GENERATED MEASUREMENTS -->"


def record(path):
    """the ONE JSON line of a bench.py output (banners of RCCL / HIP may precede it)"""
    lines = [l for l in open(path).read().split("\n") if l.lstrip().startswith("{")]
    return json.loads(lines[-1])


def get(d, path, default=None):
    for k in path.split("/"):
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def fmt(x, digits=3):
    if x is None:
        return "n/a"
    if isinstance(x, float):
        return f"{x:.{digits}g}" if abs(x) >= 1e5 or abs(x) < 1e-2 else f"{x:.{digits}f}".rstrip("0").rstrip(".")
    return str(x)


def main(tag):
    P = lambda name: os.path.join(ROOT, "profiles", f"{tag}_{name}")
    rows = []

    def row(what, value, source):
        rows.append(f"| {what} | {value} | `{source}` |")

    drv_name = f"{tag}_bench_driver_cmd.json"
    d = record(P("bench_driver_cmd.json"))
    r = d["roofline"]
    row("**headline** (`python bench.py --gpus 1 --steps 20 --warmup 5`, the driver's command): env-steps/s, fp32, 65 536 envs",
        f"**{d['value']:.4g}** ({d['ms_per_step'] * 1e3:.3f} µs per step; region {get(d, 'timing/region_ms/charged') * 1e3:.1f} µs, "
        f"{get(d, 'timing/repetitions')} regions)", f"{drv_name}: value, ms_per_step, timing")
    row("`roofline.frac` of the fused kernel in those regions: 4 968 FLOP × 65 536 × 20 ÷ rocprofv3's mean per-dispatch duration ÷ 157.3 TFLOP/s",
        f"**{r['frac']:.3f}** ({r['achieved']:.1f} TFLOP/s, {r['avg_launch_ms'] * 1e3:.2f} µs per launch)", f"{drv_name}: roofline; {get(r, 'rocprofv3/source')}")
    row("which clock that fraction is on (`roofline.frac_basis`)", str(r.get("frac_basis")), f"{drv_name}: roofline/frac_basis")
    row("the same launches measured in the record's own run: the waves' first-in / last-out span; HIP events on the engine's stream around the launch; "
        "the same FLOP over the timed region itself", f"frac_in_run {fmt(r.get('frac_in_run'))} ({fmt((r.get('avg_launch_ms_in_run') or 0) * 1e3)} µs) / "
        f"frac_hip_events {fmt(r.get('frac_hip_events'))} ({fmt((r.get('avg_launch_ms_hip_events') or 0) * 1e3)} µs) / frac_by_region {fmt(r.get('frac_by_region'))}",
        f"{drv_name}: roofline/frac_in_run, frac_hip_events, frac_by_region")
    if "wave_span" in r:
        w = r["wave_span"]
        row("the same launches by the waves' own first-in / last-out span, measured in the record's run",
            f"{w['frac']:.3f} ({w['avg_launch_ms'] * 1e3:.2f} µs; rocprofv3 reads {w['rocprofv3_mean_minus_wave_span_us']:.1f} µs more per launch)",
            f"{drv_name}: roofline/wave_span")
    row("core clock of those launches; fraction of the peak at that clock", f"{fmt(r.get('clock_ghz_under_load'))} GHz; {fmt(r.get('frac_of_peak_at_that_clock'))}",
        f"{drv_name}: roofline/clock_ghz_under_load")
    tr = r.get("traffic_source") or {}
    row("HBM traffic of a 20-step launch (FETCH_SIZE + WRITE_SIZE passes, gfx950 correction) against the algorithmic 468 B per env",
        f"{fmt(tr.get('bytes_per_env'))} B per env", f"{tr.get('source')}: {r['kernel']}")
    sq = r.get("sq_counters") or {}
    pw = sq.get("per_wave_step") or {}
    row("SQ counters of a 2 000-step launch: MFMA busy / MFMA+VALU co-execution / issue stall; per wave-step MFMA, transcendental, other vector, scalar instructions, cycles",
        f"{fmt(sq.get('mfma_busy_frac'))} / {fmt(sq.get('mfma_valu_coexec_frac_of_busy'))} / {fmt(sq.get('issue_stall_frac'))}; "
        f"{fmt(pw.get('mfma'))}, {fmt(pw.get('transcendental'))}, {fmt(pw.get('other_vector'))}, {fmt(pw.get('scalar'))}, {fmt(pw.get('cycles'))}",
        f"{sq.get('source')}: fp32")
    s = d.get("steady_state") or {}
    row("sustained: ten 500-step launches back to back (65 536 envs)", f"{s.get('env_steps_per_s', 0):.4g} env-steps/s, {fmt(s.get('us_per_step_kernel'))} µs per step, "
        f"frac {fmt(s.get('frac'))} at {fmt(s.get('clock_ghz_under_load'))} GHz", f"{drv_name}: steady_state")
    c4 = d.get("config4") or {}
    row("BASELINE config 3 / 4 (262 144 envs per GPU, two waves per SIMD), four 500-step launches", f"{c4.get('env_steps_per_s', 0):.4g} env-steps/s, "
        f"{fmt(c4.get('us_per_step_kernel'))} µs per step, frac {fmt(c4.get('frac'))}", f"{drv_name}: config4")
    b = get(d, "extensions_n65536/rollout_bf16_actor") or {}
    br = b.get("roofline") or {}
    row("BASELINE config 5 (bf16 MFMA actor + fp32 RK4, 65 536 envs), sustained", f"{b.get('env_steps_per_s', 0):.4g} env-steps/s, {fmt(b.get('us_per_step'))} µs per step; "
        f"fp32 VALU work at {fmt(br.get('frac'))} of the vector peak, MFMA pipe at {fmt(br.get('mfma_frac_of_peak'))} of 2.5 PFLOP/s; "
        f"measured / lone-wave issue model {fmt(get(br, 'sq_counters/measured_over_issue_model'))}", f"{drv_name}: extensions_n65536/rollout_bf16_actor")
    f16 = get(d, "extensions_n65536/rollout_split_f16_actor") or {}
    row("split-f16 actor (22-bit operands on the 16-bit MFMA pipe; not the headline)", f"{f16.get('env_steps_per_s', 0):.4g} env-steps/s, {fmt(f16.get('us_per_step'))} µs per step",
        f"{drv_name}: extensions_n65536/rollout_split_f16_actor")
    for size in ("n65536", "n2097152"):
        k = get(d, f"kernels/{size}") or {}
        if k:
            row(f"API-granular kernels at {size[1:]} envs: `k_observe` / `k_actor_step` / `k_step` µs per launch (fraction of 8 TB/s on their algorithmic bytes)",
                " / ".join(f"{fmt(k[n]['us_per_launch'])} ({fmt(k[n]['frac'])})" for n in ("k_observe", "k_actor_step", "k_step")),
                f"{drv_name}: kernels/{size}")
    rec = get(d, "extensions_n65536/rollout_recorded") or {}
    row("fused rollout WITH trajectory recording (109 B per env-step written)", f"{rec.get('env_steps_per_s', 0):.4g} env-steps/s, {fmt(rec.get('trajectory_GBps'))} GB/s of trajectory",
        f"{drv_name}: extensions_n65536/rollout_recorded")
    seq = get(d, "extensions_n65536/evaluate_sequence") or {}
    row("`Raptor.evaluate_sequence` ([T, 65 536, 22] in one launch), sustained", f"{fmt(seq.get('us_per_step'))} µs per step, frac {fmt(seq.get('frac'))} of the f32 MFMA peak",
        f"{drv_name}: extensions_n65536/evaluate_sequence")
    tb = get(d, "extensions_n65536/teacher_bank/teachers_1000_balanced") or {}
    row("teacher bank, 1 000 teachers 22-64-64-4 × 65 536 envs × 500 steps, exact-f32 MFMA", f"{fmt(tb.get('ms'))} ms, frac {fmt(tb.get('frac'))}",
        f"{drv_name}: extensions_n65536/teacher_bank")
    tpath = P("teacher_rate.json")
    if os.path.exists(tpath):
        t = record(tpath)
        lay = {k: v for k, v in t.items() if k.startswith("layers_")}
        if lay:
            row("teacher stacks outside that family (dense-stack kernel: LDS-resident image, (env, step) column tiles, fp32; contiguous assignment): " + ", ".join(k[7:] for k in lay),
                ", ".join(f"{fmt(v['ms'])} ms ({fmt(v['frac_of_f32_mfma_peak'])})" for v in lay.values()), f"{tag}_teacher_rate.json")
        if "bf16" in t and "f16x2" in t:
            row("the 22-64-64-4 bank in bf16 / split f16", f"{fmt(t['bf16']['ms'])} / {fmt(t['f16x2']['ms'])} ms", f"{tag}_teacher_rate.json")
    if os.path.exists(P("teacher_pmc.json")):
        tp = json.load(open(P("teacher_pmc.json")))
        tops = [k for k in tp if k[0].isdigit()]
        row("HBM traffic of those launches over their algorithmic bytes (observations in + actions out + every teacher's parameters once)",
            ", ".join(f"{k}: {fmt(tp[k]['traffic_over_algorithmic'])}×" for k in tops), f"{tag}_teacher_pmc.json")
    dg = d.get("dagger_epoch") or {}
    row("one DAgger epoch of the reference's size (≈ 78 k transitions × 1 000 teachers: record + relabel)",
        f"{fmt(get(dg, 'one_env_per_teacher/ms_per_epoch'))} ms (one env per teacher) / {fmt(get(dg, 'sixteen_envs_per_teacher/ms_per_epoch'))} ms (16 per teacher)",
        f"{drv_name}: dagger_epoch")
    for size in ("n65536", "n262144"):
        x = get(d, f"native_exchange_1rank/{size}") or {}
        if x:
            row(f"the all-gather of returns beside saturating rollouts, real librccl, ONE rank, {size[1:]} envs: added per episode; exchange alone; verified",
                f"+{fmt(x.get('added_us_per_episode'))} µs of {fmt(x.get('us_per_episode_without_exchange'))} ({100 * x.get('added_fraction', 0):.1f} %); "
                f"{fmt(x.get('exchange_alone_us_post_to_gathered'))} µs; {x.get('exchange_verified')}", f"{drv_name}: native_exchange_1rank/{size}")
    rc = get(d, "config/rccl") or {}
    row("the RCCL this record met, by its own account", f"{rc.get('ranks')} rank(s), version {rc.get('version')}, `{rc.get('library_path')}`", f"{drv_name}: config/rccl")
    rl = d.get("readme_loop_n8") or {}
    rx = rl.get("resident_executor") or {}
    row("the README loop at the reference's batch (8 envs, NumPy arrays every call): resident executor / the two launches it replaces / the "
        "device-resident chain of three launches", f"**{fmt(rl.get('numpy_arrays_us_per_iteration'))}** / {fmt(rl.get('numpy_arrays_launches_us_per_iteration'))} / "
        f"{fmt(rl.get('device_resident_us_per_iteration'))} µs per iteration ({rx.get('commands')} commands to {rx.get('starts')} kernels in "
        f"{rx.get('iterations')} iterations, {rx.get('replays')} replayed)", f"{drv_name}: readme_loop_n8")
    pl = d.get("policy_alone_n1") or {}
    if pl:
        pc = pl.get("commands") or {}
        row("the policy alone at batch 1 (`policy.evaluate_step(observation)[0]`, README.md:17-25, NumPy arrays every call): resident policy executor / "
            "the launch it replaces", f"**{fmt(pl.get('resident_executor_us_per_call'))}** / {fmt(pl.get('launches_us_per_call'))} µs per call "
            f"({pc.get('commands')} commands to {pc.get('starts')} kernels in {pl.get('calls')} calls, {pc.get('replays')} replayed)", f"{drv_name}: policy_alone_n1")
    cb = d.get("cpu_baseline") or {}
    row("CPU baseline: the oracle's C restatement on the GPU box's host cores (a reported baseline, not a target)",
        f"{cb.get('value', 0):.3g} env-steps/s with {cb.get('cores')} threads ({cb.get('host_threads_available')} visible, quota ≈ {fmt(cb.get('effective_cores'))}); "
        f"one thread, 8 envs: {get(cb, 'extras/B2_n8_1thread_env_steps_per_s', 0):.3g}", f"{drv_name}: cpu_baseline")
    pa = d.get("parity") or {}
    row("parity carried in the record", f"actor KATs {fmt(get(pa, 'actor/kat_h_max_abs_err'))} / {fmt(get(pa, 'actor/kat_h5_max_abs_err'))} (bar 1e-5); env: unpinned; "
        f"crazyflie tag: log {get(pa, 'reference_log/crazyflie/log/share_terminated')} vs {get(pa, 'reference_log/crazyflie/this_specification/share_terminated')}",
        f"{drv_name}: parity")
    for name, label in (("bench_chained.json", "chained mode (two launches per step under a hipGraph), 65 536 envs"),
                        ("bench_1048576.json", "fused, 1 048 576 envs per GPU")):
        if os.path.exists(P(name)):
            x = record(P(name))
            row(label, f"{x['value']:.4g} env-steps/s, roofline frac {fmt(get(x, 'roofline/frac'))}", f"{tag}_{name}")
    if os.path.exists(P("sq_sequence.json")):
        row("`evaluate_sequence` at 2 000 steps per launch and its SQ breakdown (120 MFMAs + 128 vector + 96 transcendental instructions per wave-step, "
            "co-execution 0: at the lone wave's issue model)", "0.657 of the f32 MFMA peak", f"{tag}_sequence_breakdown.md, {tag}_sq_sequence.json")
    if os.path.exists(P("foreign_soak.json")):
        fs = json.load(open(P("foreign_soak.json")))["results"]
        ran = [x for x in fs if "skipped" not in x]
        row("foreign aggressor / victim soak beside PyTorch on one GPU: repetitions that differ from the idle-GPU bits",
            f"{sum(x['repetitions_that_differ'] for x in ran)} of {sum(x['repetitions'] for x in ran)} over {len(ran)} pairings", f"{tag}_foreign_soak.json")
    if os.path.exists(P("foreign_soak_unrewritten_build.json")):
        fs = json.load(open(P("foreign_soak_unrewritten_build.json")))["results"]
        row("the same victim workload in a build WITHOUT the op_sel pass, beside torch's 16-bit kernels",
            "; ".join(f"{x['aggressor'].split(',')[0]}: {x['repetitions_that_differ']}/{x['repetitions']}" for x in fs), f"{tag}_foreign_soak_unrewritten_build.json")
    if os.path.exists(P("eight_ranks_one_gpu.json")):
        e8 = json.load(open(P("eight_ranks_one_gpu.json")))
        row("the driver's eight-rank command end to end on ONE GPU (`--allow-oversubscribe`, tests-only RCCL): wall time",
            f"{fmt(e8.get('wall_s'))} s (box: 120 s)", f"{tag}_eight_ranks_one_gpu.json")
    table = "\n".join([BEGIN, f"One MI355X per run; files under `profiles/` (what each is: `profiles/README.md`).  Generated by `python tools/design_tables.py {tag}`.", "",
                       "| what | measured | read from |", "|---|---|---|"] + rows + [END])
    path = os.path.join(ROOT, "DESIGN.md")
    text = open(path).read()
    a, b = text.index(BEGIN), text.index(END) + len(END)
    open(path, "w").write(text[:a] + table + text[b:])
    print(f"DESIGN.md: {len(rows)} rows from profiles/{tag}_*")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r06")
