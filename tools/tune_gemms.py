"""Record PyTorch TunableOp selections for the plain (non-batched) library GEMMs of one workload and merge the new shapes into
mmlearn_amd/tuned/gemm_gfx950.csv (look-up only at run time: mmlearn_amd.tuned.enable()).

    python tools/tune_gemms.py --workload ijepa_vitl [--steps 2] [--out gpurun_out/tuned_ijepa.csv] [--merge]

Strided-batched entries are never merged (mmlearn_amd/tuned/__init__.py: a library candidate for HTSAT's 24-wide batched products
faults).  Run on the GPU box; a shape already in the shipped file keeps its entry."""
import argparse, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", choices=("ijepa_vitl", "none"), default="ijepa_vitl")
    ap.add_argument("--untuned", default=None, help="a TunableOp record-untuned file (tools/probes/record_untuned_three_tower.py): tune exactly "
                                                     "its plain GEMM shapes offline instead of running a workload")
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "tuned_new.csv"))
    ap.add_argument("--merge", action="store_true", help="append the new plain-GEMM shapes to the shipped selections file")
    a = ap.parse_args()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    import torch
    import torch.cuda.tunable as tunable

    from mmlearn_amd import tuned

    dev = torch.device("cuda", 0)
    tunable.enable(True)
    tunable.tuning_enable(True)
    tunable.set_filename(a.out)
    tunable.read_file(tuned.DEFAULT_FILE)   # shapes already selected are not tuned again
    task = opt = None
    if a.untuned:
        lines = [ln for ln in open(a.untuned).read().splitlines() if ln.startswith(("GemmTunableOp", "GemmAndBiasTunableOp"))]
        if any("Batched" in ln for ln in open(a.untuned).read().splitlines()):
            print("[tune] the untuned file lists strided-batched ops: they are skipped", flush=True)
        tmp = a.out + ".untuned_plain.csv"
        open(tmp, "w").write("\n".join(lines) + "\n")
        print(f"[tune] tuning {len(lines)} shapes of {a.untuned}", flush=True)
        tunable.tune_gemm_in_file(tmp)
    elif a.workload == "ijepa_vitl":
        import bench_ijepa_step as T

        task = T.build(False, True, dev)
        opt = task.configure_optimizers()
        opt = opt["optimizer"] if isinstance(opt, dict) else opt
        imgs = torch.rand(128, 3, 224, 224, generator=torch.Generator().manual_seed(1)).to(dev)
        torch.manual_seed(7)
        for k in range(a.steps + 6):   # several mask geometries -> the predictor's row counts vary (4 x B x (169 + 30 .. 42))
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                loss = task.training_step({"rgb": imgs}, 0)
            loss.backward()
            opt.step()
            task.on_before_zero_grad(opt)
            torch.cuda.synchronize()
            print(f"[tune] step {k} done, loss {float(loss.detach()):.4f}", flush=True)
    del task, opt
    if hasattr(tunable, "write_file"):
        tunable.write_file(a.out)
    else:   # this torch writes the results when the tuning context goes away: flush by switching tuning off and reading them back
        tunable.tuning_enable(False)
        res = tunable.get_results()
        vals = tunable.get_validators()
        with open(a.out, "w") as f:
            for v in vals:
                f.write("Validator," + ",".join(str(x) for x in v) + "\n")
            for r in res:
                f.write(",".join(str(x) for x in r) + "\n")
    old = open(tuned.DEFAULT_FILE).read().splitlines()
    have = {ln.split(",")[0] + "," + ln.split(",")[1] for ln in old if not ln.startswith("Validator")}
    new = [ln for ln in open(a.out).read().splitlines()
           if ln.startswith(("GemmTunableOp", "GemmAndBiasTunableOp")) and (ln.split(",")[0] + "," + ln.split(",")[1]) not in have]
    print(f"[tune] {len(new)} new plain-GEMM shapes")
    for ln in new:
        print(ln)
    if a.merge and new:
        with open(tuned.DEFAULT_FILE, "a") as f:
            for ln in new:
                f.write(ln + "\n")
        print(f"[tune] merged into {tuned.DEFAULT_FILE}")


if __name__ == "__main__":
    main()
