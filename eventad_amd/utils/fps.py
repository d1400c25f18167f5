"""measure_fps of the EventAD test run (interface and definition of the reference's utils/fps.py:6-96, called from
test.py:76-83): "FPS" = boxes of the current and of the previous frame (len(bbox) + len(bbox0)) per second of evaluation
forward passes `model(data, labels, testing=True)`, each bracketed by a device synchronisation, after `warmup_batches`
untimed batches; batches without boxes are skipped and not counted.

The only number the reference publishes for its own speed is this one (595.48 on its authors' GPU,
output/test_results/.../test_results.csv:6), so the mirror keeps the definition to the letter: host clock around a
synchronised forward, boxes as the unit.  Added keys: the number of timed batches and clips per second."""
import time

import torch

from ..dagr.utils.buffers import format_data


def measure_fps(model, test_loader, device, warmup_batches=5, num_batches=20):
    device = torch.device(device)
    batch_times, batch_sizes, clips = [], [], 0
    model.eval()
    on_gpu = device.type == "cuda"
    if on_gpu:
        torch.cuda.synchronize()
    with torch.no_grad():
        for i, data in enumerate(test_loader):
            bbox = getattr(data, "bbox", None)
            if bbox is None or (isinstance(bbox, torch.Tensor) and bbox.shape[0] == 0):
                continue                                       # (the loader index still advances, as in the reference)
            data = format_data(data.to(device))
            labels = data.bbox[:, 4]
            if i < warmup_batches:
                model(data, labels, testing=True)
                continue
            if len(batch_times) >= num_batches:
                break
            bbox0 = getattr(data, "bbox0", None)
            batch_sizes.append(len(data.bbox) + (len(bbox0) if bbox0 is not None else 0))
            clips += int(getattr(data, "num_graphs", 0))
            if on_gpu:
                torch.cuda.synchronize()
            t0 = time.time()
            model(data, labels, testing=True)
            if on_gpu:
                torch.cuda.synchronize()
            batch_times.append(time.time() - t0)
    total_time, total_samples = sum(batch_times), sum(batch_sizes)
    return {"fps": total_samples / total_time if total_time > 0 else 0,
            "avg_batch_size": total_samples / len(batch_sizes) if batch_sizes else 0,
            "total_time": total_time,
            "batches": len(batch_times),
            "clips_per_second": clips / total_time if total_time > 0 else 0}
