"""parse_sample_id (reference utils/utils.py:70-93)."""
import re


def parse_sample_id(sample_id):
    """-> (None, frame_id, object_id): the numbers after 'frame_' and 'obj_' in the id, 0 when absent or when
    the id is not a string.  The video id always comes from the batch's `sequence` attribute."""
    if isinstance(sample_id, str):
        f = re.search(r'frame_(\d+)', sample_id)
        o = re.search(r'obj_(\d+)', sample_id)
        return None, (int(f.group(1)) if f else 0), (int(o.group(1)) if o else 0)
    return None, 0, 0
