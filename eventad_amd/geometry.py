"""Host-side constants of the DAGR backbone for one sensor geometry (product code; mirrors the
values the reference derives in Net.__init__ (src/dagr/model/networks/net.py:30-99),
EV_TGN.init_graph_creator (model/layers/ev_tgn.py:27-37), Pooling.__init__ (layers/pooling.py:19-35)
and DAGR.cache_luts (networks/dagr.py:36-71)).  fp32 values are produced with torch fp32 ops so they
round exactly as the reference's tensors do."""
import math

import torch


class Geometry:
    def __init__(self, width, height, radius=0.01, max_neighbors=16, pooling_dim_at_output="5x7",
                 time_window=1000000, max_queue_size=128):
        self.W, self.H, self.T = int(width), int(height), int(time_window)
        self.max_neighbors, self.Q = int(max_neighbors), int(max_queue_size)
        self.radius = radius
        self.radius_px = int(radius * width + 1)                      # ev_tgn.py:29
        self.delta_t_us = int(radius * time_window)                   # ev_tgn.py:28
        self.steps = (2 * self.radius_px + 1) ** 2
        self.eff_r = 2 * float(int(radius * width + 2) / width)       # net.py:70
        py, px = map(int, pooling_dim_at_output.split("x"))           # net.py:19-28
        base = torch.tensor([1.0 / px, 1.0 / py, 1.0])
        self.voxel = []                                               # fp32 (vx, vy) per pooling stage
        for i in range(4):
            p = base / 2 ** (3 - i)
            self.voxel.append((float(p[0]), float(p[1])))
        end = torch.tensor(0.9999999)                                 # pooling.py:31
        self.grid = [(int((end / torch.tensor(vx)).long()) + 1, int((end / torch.tensor(vy)).long()) + 1)
                     for vx, vy in self.voxel]                        # torch_cluster grid extents (nx, ny)
        mv = [2 * max(vx, vy) for vx, vy in self.voxel]               # net.py:67 (fp32 tensor * 2 is exact)
        self.cart_max = [self.eff_r, 2 * self.eff_r, mv[1], mv[2], mv[3]]   # net.py:71,75,81,87,93
        wh_inv = 1 / torch.Tensor([[width, height]])                  # pooling.py:32
        self.wh_inv = (float(wh_inv[0, 0]), float(wh_inv[0, 1]))
        # divisors of the LUT grid attribute d/(2*M*size)+0.5 (spline_conv.py:27-30), as fp32
        self.lut_div = [(float(torch.tensor(2 * m * width, dtype=torch.float32)),
                         float(torch.tensor(2 * m * height, dtype=torch.float32))) for m in self.cart_max]
        self.pool_aggr = ["max", "max", "max", "mean"]                # net.py:76-95
        for (vx, vy), (nx, ny) in zip(self.voxel[:1], self.grid[:1]):
            # pooled-edge bitmap window is 5x5 voxels: radius graph edges must stay within 2 voxels
            assert math.ceil(self.radius_px / (vx * width)) <= 2 and math.ceil(self.radius_px / (vy * height)) <= 2

    def pixel_voxel_tables(self):
        """pool1: voxel column/row of every pixel as grid_cluster computes it ((px/W)/vx truncated, fp32),
        and the first pixel of every voxel column/row (+ sentinel)."""
        vx, vy = self.voxel[0]
        nx, ny = self.grid[0]
        cx = ((torch.arange(self.W).float() / float(self.W)) / torch.tensor(vx)).long()
        cy = ((torch.arange(self.H).float() / float(self.H)) / torch.tensor(vy)).long()
        assert int(cx.max()) < nx and int(cy.max()) < ny and bool((cx[1:] >= cx[:-1]).all()) and bool((cy[1:] >= cy[:-1]).all())
        lo_x = torch.searchsorted(cx, torch.arange(nx + 1)).int()
        lo_y = torch.searchsorted(cy, torch.arange(ny + 1)).int()
        return cx.int(), cy.int(), lo_x, lo_y

    def max_nodes(self, level, B):
        nx, ny = self.grid[level]
        return nx * ny * (B + 1)
