"""Test-infrastructure stand-in for the absent third-party package `ordered_set`.

Only used by tests/golden/gen/make_golden.py, in the build container, to let the reference's
data_loader.py (data_loader.py:11,64) import. Semantics restated from the package's published
behaviour: a set that iterates in first-insertion order. Not product code.
"""


class OrderedSet(object):
    def __init__(self, items=()):
        self._pos = {}
        for it in items:
            self.add(it)

    def add(self, item):
        if item not in self._pos:
            self._pos[item] = len(self._pos)
        return self._pos[item]

    def __iter__(self):
        return iter(self._pos)

    def __len__(self):
        return len(self._pos)

    def __contains__(self, item):
        return item in self._pos
