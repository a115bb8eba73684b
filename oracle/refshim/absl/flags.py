class _Flags(object):
    def __call__(self, *a, **k):
        return None


FLAGS = _Flags()
