def scatter_mean(*a, **k):
    raise NotImplementedError("shim: torch_scatter is imported by the reference but unused on the SATrans path")
