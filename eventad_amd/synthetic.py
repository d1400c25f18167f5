"""Deterministic synthetic ROL/DoTA-shaped clips (the shapes of the benchmark configurations C1-C5).

Per clip c: CPU generator seeded 1234+c; x,y uniform over the sensor; t = N sorted draws from the
last `t_span_us` of a 1 000 000 us window with the last event forced to the window end
(src/dagr/data/dsec_data.py:127-129); polarity +-1; uint8 RGB frame; 10 tracked boxes
(ids 1..10, 20 % anomalous) with a +-2 px shifted previous-frame copy.  "Precomputed CNN
features" = the five HookModule maps drawn N(0,1) with seed 99+c.
"""
import torch

from .batch import EventBatch

FEATURE_CHANNELS = [16, 64, 64, 64, 64]        # net.py:34-37,48 with dagr-s widths
FEATURE_STRIDES = [2, 4, 8, 16, 32]            # conv1, layer1..layer4 of ResNet-50

CONFIGS = {
    # name: (W, H, events/clip, batch, live_cnn)
    "C1": (64, 64, 5000, 1, True),
    "C2": (256, 256, 50000, 8, False),
    "C3": (256, 256, 200000, 32, True),
    "C4": (1280, 720, 500000, 8, True),        # per GPU share of batch 64 over 8 GPUs
    "C5": (640, 480, 2000000, 16, False),
}


def _conv_out(n, stride):
    # spatial size of the ResNet-50 tap at this stride (7x7/2 pad 3, 3x3/2 pad 1, 3x3/2 pad 1 ...)
    s = 1
    while s < stride:
        n = (n + 1) // 2 if s >= 2 else (n + 2 * 3 - 7) // 2 + 1
        s *= 2
    return n


def feature_map_sizes(width, height):
    return [(_conv_out(height, s), _conv_out(width, s)) for s in FEATURE_STRIDES]


def make_clip(c, width, height, n_events, time_window=1000000, t_span_us=50000, n_tracks=10,
              with_image=True, precomputed_feats=False):
    g = torch.Generator().manual_seed(1234 + c)
    x = torch.randint(0, width, (n_events,), generator=g, dtype=torch.int16)
    y = torch.randint(0, height, (n_events,), generator=g, dtype=torch.int16)
    t = torch.randint(time_window - t_span_us + 1, time_window + 1, (n_events,), generator=g, dtype=torch.int32)
    t = torch.sort(t).values
    if n_events > 0:
        t[-1] = time_window
    p = (torch.randint(0, 2, (n_events,), generator=g, dtype=torch.int8) * 2 - 1).view(-1, 1)
    clip = dict(x=p, pos=torch.stack([x, y], 1), t=t, width=width, height=height, time_window=time_window,
                sequence="synthetic_%04d" % c)
    clip["image"] = (torch.randint(0, 256, (1, 3, height, width), generator=g, dtype=torch.uint8)
                     if with_image else None)
    lo, hi = 16, max(17, width // 4)
    w = torch.randint(lo, hi + 1, (n_tracks,), generator=g).float()
    h = torch.randint(lo, hi + 1, (n_tracks,), generator=g).float()
    bx = torch.rand(n_tracks, generator=g) * (width - w).clamp(min=0)
    by = torch.rand(n_tracks, generator=g) * (height - h).clamp(min=0)
    cls = (torch.rand(n_tracks, generator=g) < 0.2).float()
    tid = torch.arange(1, n_tracks + 1).float()
    clip["bbox"] = torch.stack([bx, by, w, h, cls, tid], 1)
    shift = torch.randint(-2, 3, (n_tracks, 2), generator=g).float()
    clip["bbox0"] = torch.stack([bx + shift[:, 0], by + shift[:, 1], w, h, cls, tid], 1)
    clip["image_feats"] = None
    if precomputed_feats:
        gf = torch.Generator().manual_seed(99 + c)
        clip["image_feats"] = [torch.randn(1, ch, hh, ww, generator=gf)
                               for ch, (hh, ww) in zip(FEATURE_CHANNELS, feature_map_sizes(width, height))]
    return clip


def make_batch(width, height, n_events, batch_size, first_clip=0, with_image=True,
               precomputed_feats=False, **kw):
    clips = [make_clip(first_clip + i, width, height, n_events, with_image=with_image,
                       precomputed_feats=precomputed_feats, **kw) for i in range(batch_size)]
    return EventBatch.from_clips(clips)


def make_config_batch(name, first_clip=0, batch_size=None, n_events=None):
    W, H, n, B, live = CONFIGS[name]
    return make_batch(W, H, n_events or n, batch_size or B, first_clip=first_clip, with_image=live,
                      precomputed_feats=not live)
