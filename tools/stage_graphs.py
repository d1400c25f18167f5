"""Time the two pipeline stages of the C2 step as separate hipGraphs (backbone + ROI pooling; head update) and
the combined pipelined graph: shows how much of the overlap the two streams actually achieve."""
import sys, time, torch
sys.path.insert(0, ".")
import bench

def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
    dev = torch.device("cuda:0")
    from eventad_amd import synthetic
    from eventad_amd.train_step import TrainStep
    W, H, n_ev, B, live = synthetic.CONFIGS[cfg]
    args, model, sd = bench.build_model(W, H, B, dev, live_cnn=live)
    from dagr.utils.buffers import format_data
    data = format_data(synthetic.make_config_batch(cfg).clone().to(dev))
    labels = data.bbox[:, 4]
    model.train()
    step = TrainStep(model, lr=args.learning_rate, weight_decay=1e-5, grad_clip=args.grad_clip)
    for _ in range(3):
        step.step(data)
    step.prime(data)
    keep = []
    def timeit(body, n=200):
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3): body()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        keep.append(g)                      # graphs (and their private pools) stay alive for the whole run
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            body()
        for _ in range(10): g.replay()
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(n): g.replay()
        cpu = (time.perf_counter() - t) / n * 1e3               # host time to enqueue one replay
        torch.cuda.synchronize()
        print("  host enqueue %.3f ms/replay" % cpu)
        return (time.perf_counter() - t) / n * 1e3
    def bb():
        with torch.no_grad():
            model.extract_features(data, frames=(1,))
    def head():
        with torch.no_grad():
            model._head_native(step._feats_cur, data, labels)
            step.nh.backward()
            step.nh.optimizer_step()
    which = sys.argv[2] if len(sys.argv) > 2 else "backbone"     # ONE capture per process (several large graphs in
    if which == "backbone":                                      # one process have crashed the HIP runtime here)
        print("backbone + ROI graph : %.3f ms" % timeit(bb))
    elif which == "head":
        print("head update graph    : %.3f ms" % timeit(head))
    elif which == "sequential":
        print("sequential step graph: %.3f ms" % timeit(lambda: step.step(data, labels)))
    else:
        print("pipelined step graph : %.3f ms" % timeit(lambda: step.pipelined_step(data, labels, data)))

if __name__ == "__main__":
    main()
