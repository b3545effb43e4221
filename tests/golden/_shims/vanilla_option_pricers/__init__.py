def __getattr__(name):
    if name.startswith("__"):
        raise AttributeError(name)

    def _missing(*a, **k):
        raise NotImplementedError(f"vanilla_option_pricers.{name} is not available in this container")
    return _missing
