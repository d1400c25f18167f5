import os, sys, json, torch, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import build_model
from eventad_amd import synthetic
from eventad_amd.dagr.utils.buffers import format_data
W = H = 256; B = 6; n = 50000
dev = torch.device("cuda:0")
args, model, _ = build_model(W, H, B, dev, live_cnn=True)
model.eval()
batches = [format_data(synthetic.make_batch(W, H, n, B, first_clip=i * B, with_image=True).to(dev)) for i in range(8)]
with torch.no_grad():
    for d in batches[:4]:
        model(d, d.bbox[:, 4], testing=True)
    torch.cuda.synchronize()
    # host time to issue (no sync) vs time to finish
    t0 = time.time()
    for d in batches[4:]:
        model(d, d.bbox[:, 4], testing=True)
    t1 = time.time()
    torch.cuda.synchronize()
    t2 = time.time()
    print("issue %.2f ms per batch, finish %.2f ms per batch" % ((t1 - t0) * 250, (t2 - t0) * 250))
    pr = cProfile.Profile()
    pr.enable()
    for d in batches[4:]:
        model(d, d.bbox[:, 4], testing=True)
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
