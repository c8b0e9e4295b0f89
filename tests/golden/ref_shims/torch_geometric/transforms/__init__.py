class Compose:
    def __init__(self, ts):
        self.ts = ts

    def __call__(self, d):
        for t in self.ts:
            d = t(d)
        return d
