"""Pieces of `dagr.asynchronous` (reference src/dagr/asynchronous/): the native index-masked row kernels
(`asy_tools`).  The event-by-event rewrite of the layers themselves (make_model_asynchronous, __init__.py:41-110) is not
implemented."""
