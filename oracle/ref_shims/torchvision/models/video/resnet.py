def r2plus1d_18(*a, **k):
    """Placeholder: the reference builds this backbone in VideoEncoder.__init__ (model.py:153) and never uses it."""
    return None
