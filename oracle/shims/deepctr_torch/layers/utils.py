def slice_arrays(arrays, start=None, stop=None):
    if arrays is None:
        return [None]
    if isinstance(arrays, list):
        return [None if a is None else a[start:stop] for a in arrays]
    return arrays[start:stop]
