"""DyNCA model family (reference: ConditioneDyNCA/models, ExtraChannels/models).  Unlike the reference's
models/__init__.py this package fetches nothing from the network."""
