def __getattr__(name):
    raise AttributeError(name)
