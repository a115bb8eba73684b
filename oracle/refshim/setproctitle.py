def setproctitle(*a, **k):
    pass
