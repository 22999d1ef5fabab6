"""``TrackerParams`` attribute bag (lib/test/utils/params.py:5-25)."""


class TrackerParams:
    def set_default_values(self, default_vals: dict):
        for name, val in default_vals.items():
            if not hasattr(self, name):
                setattr(self, name, val)

    def get(self, name: str, *default):
        if len(default) > 1:
            raise ValueError("Can only give one default value.")
        return getattr(self, name, *default)

    def has(self, name: str):
        return hasattr(self, name)
