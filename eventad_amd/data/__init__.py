"""Input side of the path: event windows in the `events_2x.h5` schema -> per-clip tensors -> pinned batch."""
