def set_trace(*a, **k):
    raise RuntimeError("ipdb.set_trace stub reached")
