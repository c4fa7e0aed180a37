"""Name -> class registries for predictors / correctors / SDEs (same role and public methods as the
reference's utils/registry.py:5-36: register(name) decorator, get_by_name, get_all_names)."""


class Registry:
    def __init__(self, kind):
        self.kind = kind
        self._items = {}

    def register(self, name):
        def deco(cls):
            self._items[name] = cls
            return cls
        return deco

    def get_by_name(self, name):
        try:
            return self._items[name]
        except KeyError:
            raise ValueError(f"{self.kind} with name '{name}' unknown.") from None

    def get_all_names(self):
        return list(self._items)
