"""The fused stem (conv1 statistics + XL_OP_STEM12) alone on random data, HIP events; XL_STEM12_CLK=1 prints per-phase shader clocks.
usage: python tools/stem12_bench.py [frames]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from crossloc_amd import networks, synth  # noqa: E402
from crossloc_amd.weights import seeded_state_dict  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 47
    net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
    net.load_state_dict(seeded_state_dict(net, seed=2021))
    net = net.cuda().eval()
    x = torch.rand(B, 3, 480, 720, device="cuda")
    st = torch.cuda.Stream()
    with torch.cuda.stream(st), torch.no_grad():
        net(x)
        plan = list(net._plans.values())[0]
        idx = [i for i, op in enumerate(plan.ops) if op.type == networks.XL_OP_STEM12]
        if not idx:
            print("no XL_OP_STEM12 in the plan")
            return
        arr = (networks.XlOp * 1)(plan.op_array[idx[0]])
        import ctypes
        L = networks._bind()
        s = ctypes.c_void_p(st.cuda_stream)
        for _ in range(2):
            networks._check(L.xl_cnn_run(arr, 1, s))
        torch.cuda.synchronize()
        if os.environ.get("XL_STEM12_CLK"):
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            L.xl_cnn_run(arr, 1, s)
        e1.record()
        torch.cuda.synchronize()
        print("stem12 %d frames: %.4f ms" % (B, e0.elapsed_time(e1) / 10))


if __name__ == "__main__":
    main()
