"""A 1x1 512->512 layer of the 95-frame plan alone, with and without its statistics epilogue, HIP events.
usage: python tools/conv1x1_stats_bench.py [frames]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from crossloc_amd import networks, synth
from crossloc_amd.weights import seeded_state_dict
B = int(sys.argv[1]) if len(sys.argv) > 1 else 95
net = networks.TransPoseNet(torch.tensor(synth.SCENE_MEAN, dtype=torch.float32), False, False, 2, 2, 3, 1)
net.load_state_dict(seeded_state_dict(net, seed=2021))
net = net.cuda().eval()
x = torch.rand(B, 3, 480, 720, device="cuda")
st = torch.cuda.Stream()
with torch.cuda.stream(st), torch.no_grad():
    net(x)
    plan = list(net._plans.values())[0]
    L = networks._bind()
    s = ctypes.c_void_p(st.cuda_stream)
    for want_norm in (True, False):
        idx = [i for i, op in enumerate(plan.op_array) if op.type == networks.XL_OP_CONV and op.ksize == 1 and op.nchunks2 <= 1
               and op.Cin == 512 and op.Cout == 512 and bool(op.flags & networks.CONV_NORM_IN) == want_norm and op.stats
               and not (op.flags & networks.CONV_NORM_ADD)]
        if not idx:
            continue
        for with_stats in (True, False, True, False):
            arr = (networks.XlOp * 1)(plan.op_array[idx[0]])
            if not with_stats:
                arr[0].stats = None
            for _ in range(3):
                networks._check(L.xl_cnn_run(arr, 1, s))
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                L.xl_cnn_run(arr, 1, s)
            e1.record()
            torch.cuda.synchronize()
            print("1x1 512->512 norm=%s stats=%s: %.4f ms" % (want_norm, with_stats, e0.elapsed_time(e1) / 20))
