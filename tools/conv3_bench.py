"""conv3 (XL_OP_CONV 64 -> 128, stride 2, split pipe) alone out of a 95-frame plan, HIP events. usage: python tools/conv3_bench.py [frames] [cout]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossloc_amd import networks, synth
from crossloc_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 95
cout = int(sys.argv[2]) if len(sys.argv) > 2 else 128
net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
x = torch.rand(B, 3, 480, 720, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st), torch.no_grad():
    net(x)
    plan = list(net._plans.values())[0]
    idx = [i for i, op in enumerate(plan.ops) if op.type == networks.XL_OP_CONV and op.ksize == 3 and op.stride == 2 and op.Cout == cout]
    arr = (networks.XlOp * 1)(plan.op_array[idx[0]])
    L = networks._bind()
    s = ctypes.c_void_p(st.cuda_stream)
    for _ in range(2):
        networks._check(L.xl_cnn_run(arr, 1, s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        L.xl_cnn_run(arr, 1, s)
    e1.record()
    torch.cuda.synchronize()
    print("conv %d->%d, %d frames, stats %s: %.4f ms" % (arr[0].Cin, cout, B, bool(arr[0].stats), e0.elapsed_time(e1) / 10))
