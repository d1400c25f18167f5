/*
 * oracle/graph_ref.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement, in plain C, of the three kernels of the reference's native
 * graph builder.  Each function cites the reference lines it follows; paths are
 * relative to /root/reference.
 *
 *   ora_insert_in_queue         src/dagr/graph/ev_graph.cu:169-212
 *   ora_insert_in_queue_single  src/dagr/graph/ev_graph.cu:130-166
 *   ora_fill_edges              src/dagr/graph/ev_graph.cu:15-80  (+ spiral.h:1-15)
 *
 * The reference runs one GPU thread per active pixel / per event; here the same
 * per-thread bodies run in a serial loop.  The algorithm is order independent
 * across threads (each thread owns one FIFO column / one row of 16 edge slots),
 * so the serial loop gives the reference's result bit for bit.
 *
 * PARITY UNPINNED w.r.t. reference outputs: the reference holds no tests or golden
 * vectors for this path and cannot be compiled here (needs CUDA + a GPU); this file
 * is pinned only by the hand-derived known-answer cases in tests/test_oracle_graph.py.
 */
#include <stdint.h>

/* spiral.h:1-15 -- square spiral (0,0),(1,0),(1,1),(0,1),(-1,1),(-1,0),... */
typedef struct { unsigned layer, leg; int x, y; } spiral_t;

static void spiral_init(spiral_t* s) { s->layer = 1; s->leg = 0; s->x = 0; s->y = 0; }

static void spiral_next(spiral_t* s) {
    switch (s->leg) {
    case 0: ++s->x; if (s->x == (int)s->layer) ++s->leg; break;
    case 1: ++s->y; if (s->y == (int)s->layer) ++s->leg; break;
    case 2: --s->x; if (-s->x == (int)s->layer) ++s->leg; break;
    case 3: --s->y; if (-s->y == (int)s->layer) { s->leg = 0; ++s->layer; } break;
    }
}

/* Exported so tests can check the spiral order itself (known answers). */
void ora_spiral(int n, int32_t* xy) {
    spiral_t s; spiral_init(&s);
    for (int i = 0; i < n; ++i) { xy[2 * i] = s.x; xy[2 * i + 1] = s.y; spiral_next(&s); }
}

/* ev_graph.cu:169-212.  One iteration of the outer loop = one reference thread. */
void ora_insert_in_queue(const int32_t* indices, const int32_t* unique_coords,
                         const int32_t* cumsum_counts, int32_t* queue,
                         int B, int Q, int H, int W, int K) {
    (void)B;
    for (int lin = 0; lin < K; ++lin) {
        int counts, offset;
        if (lin > 0) { offset = cumsum_counts[lin - 1]; counts = cumsum_counts[lin] - offset; }
        else         { offset = 0;                      counts = cumsum_counts[lin]; }
        int x = unique_coords[lin] % W;
        int y = ((unique_coords[lin] - x) / W) % H;
        int b = unique_coords[lin] / (W * H);
        for (int q = Q - 1; q >= 0; --q) {
            int64_t index = (int64_t)b * H * W * Q + (int64_t)q * H * W + y * W + x;
            if (q >= counts) {
                int64_t shifted = (int64_t)b * H * W * Q + (int64_t)(q - counts) * H * W + y * W + x;
                queue[index] = queue[shifted];
            } else {
                queue[index] = indices[offset + counts - 1 - q];
            }
        }
    }
}

/* ev_graph.cu:130-166: exactly one event, clip index forced to 0. */
void ora_insert_in_queue_single(const int32_t* indices, const int32_t* events,
                                int32_t* queue, int B, int Q, int H, int W) {
    (void)B;
    int counts = 1, offset = 0;
    int x = events[0], y = events[1], b = 0;
    for (int q = Q - 1; q >= 0; --q) {
        int64_t index = (int64_t)b * H * W * Q + (int64_t)q * H * W + y * W + x;
        if (q >= counts) {
            int64_t shifted = (int64_t)b * H * W * Q + (int64_t)(q - counts) * H * W + y * W + x;
            queue[index] = queue[shifted];
        } else {
            queue[index] = indices[offset + counts - 1 - q];
        }
    }
}

/* ev_graph.cu:15-80.  `edges` is int64[2][K], prefilled with -1 by the caller
 * (ev_graph.py:89); row 0 = source, row 1 = destination. */
void ora_fill_edges(const int32_t* batch, const int32_t* pos, const int32_t* all_timestamps,
                    const int32_t* indices, const int32_t* event_queue, int64_t* edges,
                    int B, int Q, int H, int W, int N, int64_t K, float radius,
                    float delta_t_us, int max_num_neighbors, int min_index) {
    (void)B;
    for (int ev = 0; ev < N; ++ev) {
        int radius_int = (int)radius;
        int num_neighbors = 0;
        int64_t offset = (int64_t)ev * max_num_neighbors;
        int b = batch[ev];
        int x = pos[3 * ev + 0], y = pos[3 * ev + 1], ts_event = pos[3 * ev + 2];

        edges[offset + num_neighbors + K * 0] = indices[ev] - min_index;   /* self edge first */
        edges[offset + num_neighbors + K * 1] = indices[ev] - min_index;
        num_neighbors++;

        spiral_t sp; spiral_init(&sp);
        int steps = (2 * radius_int + 1) * (2 * radius_int + 1);
        for (int i = 0; i < steps; ++i) {
            if (num_neighbors >= max_num_neighbors) break;
            for (int q = 0; q < Q; ++q) {
                int xn = x + sp.x, yn = y + sp.y;
                if (!((xn >= 0) && (yn >= 0) && (xn < W) && (yn < H))) break;     /* out of FOV */
                int64_t qi = xn + (int64_t)W * yn + (int64_t)H * W * q + (int64_t)H * W * Q * b;
                int idx = event_queue[qi];
                if (idx < min_index) break;                                       /* FIFO exhausted */
                if (indices[ev] > idx) {
                    int32_t dt_us = ts_event - all_timestamps[idx - min_index];
                    if ((float)dt_us > delta_t_us) continue;                      /* too old: keep scanning */
                    edges[offset + num_neighbors + K * 0] = idx - min_index;
                    edges[offset + num_neighbors + K * 1] = indices[ev] - min_index;
                    num_neighbors++;
                    if (num_neighbors >= max_num_neighbors) break;
                }
            }
            spiral_next(&sp);
        }
    }
}
