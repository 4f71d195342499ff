"""Round 6 (VERDICT r5 "next" #3): where does the producer / consumer X3 conv (cin >= 64) lose against the single-role kernel?  Per layer shape of
the inference volumes (192^2 input: 48^2 / 24^2 / 12^2 at n = 10 / 40) and of the training step (256^2: 64^2 / 32^2 / 16^2 at n = 16 / 32), both forms
(-DCTL_TUNING build, CTL_X3_PC = 0 / 1 in a child process each).  CTL_TOOL_LIB=tuning python tools/r6_pc_gate.py"""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(64, 64, 48), (64, 128, 24), (128, 128, 24), (128, 128, 12), (128, 64, 24), (64, 32, 48), (64, 64, 24),      # inference, 192^2 input
          (64, 64, 64), (128, 128, 32), (128, 128, 16), (64, 128, 32), (128, 64, 32), (64, 32, 64), (64, 64, 32)]      # training, 256^2 input
NS = {48: (10, 40), 24: (10, 40), 12: (10, 40), 64: (16, 32), 32: (16, 32), 16: (16, 32)}


def child():
    import torch
    from _variant import use_variant
    _ffi = use_variant()
    from cooperative_training_and_latent_space_data_augmentation_amd import ops
    from cooperative_training_and_latent_space_data_augmentation_amd._ffi import lib, check
    res = {}
    for cin, cout, h in SHAPES:
        for n in NS[h]:
            for stats in (0, 1):
                x = torch.randn(n, cin, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
                w = torch.randn(cout, cin, 3, 3, device="cuda") * 0.1
                sc, sh = torch.rand(cin, device="cuda") + 0.5, torch.randn(cin, device="cuda") * 0.1
                d = _ffi.conv_desc(n=n, hin=h, win=h, cin=cin, hout=h, wout=h, cout=cout, ks=3, pro_affine=1, pro_slope=0.2,
                                   epi_flags=_ffi.EPI_BIAS | (_ffi.EPI_STATS if stats else 0), dt=_ffi.DT_X3)
                b = torch.zeros(cout, device="cuda")
                wp = ops.pack_oihw_fwd_x3(w)
                y = torch.empty(n, cout, h, h, device="cuda").contiguous(memory_format=torch.channels_last)
                st = torch.empty(max(lib.ctl_conv_stats_floats(_ffi.desc_ptr(d)), 1), device="cuda")
                run = lambda: check(lib.ctl_conv_forward(_ffi.desc_ptr(d), x.data_ptr(), wp.data_ptr(), b.data_ptr(), sc.data_ptr(), sh.data_ptr(), None, None,
                                                         None, y.data_ptr(), st.data_ptr() if stats else None, ops.stream_ptr()))
                for _ in range(5):
                    run()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(50):
                    run()
                e1.record()
                torch.cuda.synchronize()
                res[f"{cin}->{cout}@{h} n{n} {'train' if stats else 'eval'}"] = round(e0.elapsed_time(e1) * 1e3 / 50, 2)
    print("RESULT " + json.dumps(res))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child()
    else:
        out = {}
        for pc in ("0", "1"):
            env = dict(os.environ, CTL_X3_PC=pc, CTL_TOOL_LIB=os.environ.get("CTL_TOOL_LIB", "tuning"))
            r = subprocess.run([sys.executable, __file__, "child"], capture_output=True, text=True, env=env)
            line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")]
            if not line:
                print(r.stderr[-2000:])
                raise SystemExit(1)
            out[pc] = json.loads(line[0][7:])
        print(f"{'layer':34s} {'single-role us':>14s} {'prod/cons us':>13s}  ratio")
        for k in out["0"]:
            a, b = out["0"][k], out["1"][k]
            print(f"{k:34s} {a:14.2f} {b:13.2f}  {b / a:5.2f}{'   <- PC loses' if b > 1.02 * a else ''}")
