def second2DHM(seconds):
    """seconds -> 'Dd Hh Mm' string (itr/metricmodule/__init__.py)."""
    m, s = divmod(int(seconds), 60)
    h, m = divmod(m, 60)
    d, h = divmod(h, 24)
    return "%dd %dh %dm" % (d, h, m)
