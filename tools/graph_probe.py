"""Feasibility probe: capture pieces of / one whole cooperative step in a HIP graph via torch.cuda.graph and time the replay
against the eager step.  Stages run in child processes (a crash in one does not hide the others); faulthandler prints the
Python stack of a segfault."""
import faulthandler, os, subprocess, sys, time
faulthandler.enable()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

STAGES = ["elem", "net_fwd", "net_fwd_bwd", "dual_fwd_bwd", "dropout", "mask", "adam", "std_fwd", "std_fwd_bwd", "step_noDA", "step_noopt",
          "step_one_stream", "step_two_streams"]


def timeit(fn, n=20):
    import torch
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def capture(fn, warm=3, solver=None):
    import torch
    if solver is not None:
        # tensors the solver keeps (z_i, z_s, latent_code) hold the previous call's autograd graph alive, and with it the AccumulateGrad
        # nodes of the flat parameters -- created on the stream of the FIRST call (the default stream).  Under capture they would pull
        # the legacy stream into the capture (hip::Stream::EndCapture then segfaults): drop them so that the warm-up on the capture
        # stream re-creates them there.
        solver.z_i = solver.z_s = None
        solver.latent_code = {"image": None, "segmentation": None, "shape": None}
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(cap):
        for _ in range(warm): fn()
    torch.cuda.current_stream().wait_stream(cap)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    t0 = time.perf_counter()
    with torch.cuda.graph(g, stream=cap):
        out = fn()
    torch.cuda.synchronize()
    print(f"  captured in {time.perf_counter() - t0:.2f} s", flush=True)
    return g, out


def child(stage):
    import torch
    from cooperative_training_and_latent_space_data_augmentation_amd import ops
    from cooperative_training_and_latent_space_data_augmentation_amd.solver import AdvancedTripletReconSegmentationModel
    import bench
    torch.manual_seed(0)
    s = AdvancedTripletReconSegmentationModel(use_gpu=True)
    clean, label, noisy, _ = bench.synthetic(16, 256, 256, 1000, "cuda")
    cfg = (bench.DROP_IMG, bench.DROP_SEG)
    if os.environ.get("PROBE_TARGETED") == "1":
        cfg = ({"loss_name": "mse", "mask_type": "channel", "max_threshold": 0.5, "random_threshold": False, "if_soft": True},
               {"loss_name": "ce", "mask_type": "spatial", "max_threshold": 0.5, "random_threshold": False, "if_soft": True})
    if stage == "elem":
        x = torch.randn(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        fn = lambda: ops.softmax_t_fwd(x, 2.0)
    elif stage == "net_fwd":
        net = s.model["shape_encoder"]
        x = torch.rand(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        def fn():
            with torch.no_grad():
                return net(x)
    elif stage == "net_fwd_bwd":
        net = s.model["shape_encoder"]
        x = torch.rand(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        def fn():
            net.zero_grad()
            z = net(x)
            z.backward(torch.ones_like(z))
            return z
    elif stage.startswith("nfb:"):
        net = s.model[stage[4:]]
        x = noisy if stage[4:] == "image_encoder" else (torch.rand(16, 4, 256, 256, device="cuda") if stage[4:] == "shape_encoder"
                                                        else torch.rand(16, 128, 16, 16, device="cuda"))
        x = x.contiguous(memory_format=torch.channels_last)
        def fn():
            net.zero_grad()
            y = net(x)
            y = y[0] + y[1] if isinstance(y, tuple) else y
            y.backward(torch.ones_like(y))
            return y
    elif stage == "loss_ce":
        from cooperative_training_and_latent_space_data_augmentation_amd.autograd import cross_entropy_2D
        x = torch.randn(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        def fn():
            x.grad = None
            l = cross_entropy_2D(x, label)
            l.backward()
            return l
    elif stage == "loss_mse":
        from cooperative_training_and_latent_space_data_augmentation_amd.autograd import scaled_mse
        x = torch.randn(16, 1, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        def fn():
            x.grad = None
            l = scaled_mse(x, clean, 0.5)
            l.backward()
            return l
    elif stage == "stn_pair":
        def fn():
            s.reset_all_optimizers()
            x = torch.randn(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
            a, b = s.recon_shape_pair(label, True, x, False)
            (a.sum() + b.sum()).backward()
            return a
    elif stage == "stn_pair_fixed":
        x = torch.randn(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        def fn():
            s.reset_all_optimizers()
            a, b = s.recon_shape_pair(label, True, x, False)
            (a.sum() + b.sum()).backward()
            return a
    elif stage == "stn_pair_fwd":
        x = torch.randn(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        def fn():
            with torch.no_grad():
                return s.recon_shape_pair(label, True, x, False)[0]
    elif stage == "grouped_enc":
        from cooperative_training_and_latent_space_data_augmentation_amd.autograd import net_apply
        x = torch.rand(32, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        net = s.model["shape_encoder"]
        def fn():
            net.zero_grad()
            y = net_apply(net, x, groups=2)[0]
            y.backward(torch.ones_like(y))
            return y
    elif stage == "grouped_dec":
        from cooperative_training_and_latent_space_data_augmentation_amd.autograd import net_apply
        x = torch.rand(32, 128, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        net = s.model["shape_decoder"]
        def fn():
            net.zero_grad()
            y = net_apply(net, x, groups=2)[0]
            y.backward(torch.ones_like(y))
            return y
    elif stage == "cat_split":
        from cooperative_training_and_latent_space_data_augmentation_amd.autograd import split_halves
        a0 = torch.rand(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        b0 = torch.rand(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_(True)
        def fn():
            a0.grad = b0.grad = None
            u, v = split_halves(torch.cat([a0, b0], 0) * 2.0)
            (u.sum() + 2 * v.sum()).backward()
            return u
    elif stage == "std_nogroup":
        s.two_streams = False
        s.group_stn_passes = False
        def fn():
            s.reset_all_optimizers()
            l = s.standard_training(clean, label, noisy)
            (l[0] + l[1] + l[2] + l[3]).backward()
            return l
    elif stage in ("two_chains", "two_chains_serial"):
        # two independent launch chains (no cross edges between fork and join): does graph replay overlap them like two eager streams?
        na, nb = s.model["shape_encoder"], s.model["image_decoder"]
        xa = torch.rand(16, 4, 256, 256, device="cuda").contiguous(memory_format=torch.channels_last)
        xb = torch.rand(16, 128, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        side = torch.cuda.Stream()
        def chain(net, x, reps=3):
            for _ in range(reps):
                net.zero_grad()
                y = net(x)
                y.backward(torch.ones_like(y))
        def fn():
            cur = torch.cuda.current_stream()
            if stage == "two_chains":
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    chain(nb, xb)
                chain(na, xa)
                cur.wait_stream(side)
            else:
                chain(nb, xb)
                chain(na, xa)
    elif stage == "two_nets":
        def fn():
            s.reset_all_optimizers()
            zi, zs = s.model["image_encoder"](noisy)
            y = s.model["segmentation_decoder"](zs)
            y.backward(torch.ones_like(y))
            return y
    elif stage == "dual_fwd_bwd":
        net = s.model["image_encoder"]
        def fn():
            net.zero_grad()
            zi, zs = net(noisy)
            (zi.sum() + zs.sum()).backward()
            return zi
    elif stage == "dropout":
        z = torch.rand(16, 128, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        fn = lambda: ops.dropout2d(z, 0.5, seed=5)
    elif stage == "mask":
        z = torch.rand(16, 128, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        gr = torch.randn(16, 128, 16, 16, device="cuda").contiguous(memory_format=torch.channels_last)
        fn = lambda: ops.latent_mask_apply(z, ops.latent_score(gr, 0), 0, 40)
    elif stage == "adam":
        fn = lambda: s.optimize_all_params()
    elif stage == "std_fwd":
        s.two_streams = False
        def fn():
            with torch.no_grad():
                return s.standard_training(clean, label, noisy)
    elif stage == "std_fwd_bwd":
        s.two_streams = False
        def fn():
            s.reset_all_optimizers()
            l = s.standard_training(clean, label, noisy)
            (l[0] + l[1] + l[2] + l[3]).backward()
            return l
    elif stage == "step_noDA":
        s.two_streams = False
        fn = lambda: s.cooperative_step(clean, label, noisy, *cfg, latent_DA=False)
    elif stage == "step_noopt":
        s.two_streams = False
        fn = lambda: s.cooperative_step(clean, label, noisy, *cfg, do_optim=False)
    else:
        s.two_streams = stage == "step_two_streams"
        fn = lambda: s.cooperative_step(clean, label, noisy, *cfg)
    for _ in range(3): fn()
    print(f"{stage}: eager {timeit(fn):.3f} ms", flush=True)
    g, out = capture(fn, solver=s)
    for _ in range(3): g.replay()
    print(f"{stage}: graph replay {timeit(g.replay):.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    for st in (sys.argv[1:] or STAGES):
        r = subprocess.run([sys.executable, __file__, "child", st], capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(f"{st}: FAILED rc={r.returncode}\n" + "\n".join(l for l in r.stderr.splitlines() if "Warning" not in l and "amdgpu.ids" not in l)[-3000:], flush=True)
