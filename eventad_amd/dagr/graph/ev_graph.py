"""Mirror of `dagr.graph.ev_graph.SlidingWindowGraph` (reference src/dagr/graph/ev_graph.py:18-166)
for callers that drive the graph builder themselves: same constructor, `reset()`, and
`forward(batch, pos, delete_nodes=False, collect_edges=True)` returning edge_index.  State (growing
event ids, timestamps, the dense FIFO tensor) lives in this object exactly as in the reference; the
kernels are the reference-compatible entry points of the C ABI (group A of include/eventad_hip.h).
The EventAD hot path itself does not go through this class (it uses the FIFO-free native builder)."""
import torch

from ..._lib import call, ptr


class SlidingWindowGraph:
    def __init__(self, width=640, height=480, batch_size=1, max_num_neighbors=16, max_queue_size=1024, radius=7,
                 delta_t_us=600000):
        self.width, self.height, self.batch_size = width, height, batch_size
        self.max_num_neighbors, self.max_queue_size = max_num_neighbors, max_queue_size
        self.radius, self.delta_t_us = radius, delta_t_us
        self.device = None
        self.event_queue = self.edge_buffer = None
        self.reset()

    def reset(self):
        self.edges = torch.zeros((2, 0), dtype=torch.long, device=self.device)
        self.all_timestamps = torch.zeros((0,), dtype=torch.int32, device=self.device)
        self.max_index = self.min_index = 0
        if self.event_queue is not None:
            self.event_queue.fill_(-1)

    def forward(self, batch, pos, delete_nodes=False, collect_edges=True):
        if delete_nodes:
            raise NotImplementedError("delete_nodes=True is the asynchronous-inference mode (out of scope)")
        n = len(batch)
        if self.device is None:
            self.device = batch.device
            self.reset()
            self.event_queue = torch.full((self.batch_size, self.max_queue_size, self.height, self.width), -1,
                                          dtype=torch.int32, device=self.device)
        if n == 0:
            return torch.zeros((2, 0), device=self.device, dtype=torch.int32)
        assert batch.dtype == torch.int32, batch.dtype
        st = torch.cuda.current_stream().cuda_stream
        B, Q, H, W = self.event_queue.shape
        pos = pos.int().contiguous()
        self.all_timestamps = torch.cat([self.all_timestamps, pos[:, 2]]).contiguous()
        indices = (self.max_index + torch.arange(n, dtype=torch.int32, device=self.device)).contiguous()
        self.max_index += n
        if n > 1:
            lin = pos[:, 0] + W * pos[:, 1] + W * H * batch
            s_lin, s_idx = torch.sort(lin, stable=True)
            uniq, counts = torch.unique_consecutive(s_lin, return_counts=True)
            # named tensors: a temporary would be freed -- and its block handed to the next temporary -- before the launch
            sorted_ids, coords, ends = indices[s_idx].contiguous(), uniq.int().contiguous(), torch.cumsum(counts, 0).int().contiguous()
            call("eventad_insert_in_queue", ptr(sorted_ids), ptr(coords), ptr(ends), ptr(self.event_queue), B, Q, H, W,
                 len(uniq), st)
        else:
            flat = pos.reshape(-1)
            call("eventad_insert_in_queue_single", ptr(indices), ptr(flat), ptr(self.event_queue), B, Q, H, W, st)
        K = self.max_num_neighbors * n
        if self.edge_buffer is None or self.edge_buffer.shape[1] < K:
            self.edge_buffer = torch.empty((2, K), dtype=torch.int64, device=self.device)
        buf = self.edge_buffer[:, :K].contiguous() if self.edge_buffer.shape[1] != K else self.edge_buffer
        buf.fill_(-1)
        batch = batch.contiguous()
        call("eventad_fill_edges", ptr(batch), ptr(pos), ptr(self.all_timestamps), ptr(self.event_queue),
             ptr(indices), self.max_num_neighbors, float(self.radius), float(self.delta_t_us), ptr(buf),
             self.min_index, B, Q, H, W, n, K, st)
        edges = buf[:, buf[1] >= 0]
        if collect_edges:
            self.edges = torch.cat([self.edges, edges], dim=-1)
        return edges


AsyncGraph = SlidingWindowGraph
