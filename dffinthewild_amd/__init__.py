"""dffinthewild_amd — MI355X-native (gfx950) inference path for the depth-from-focus network of
"Learning Depth from Focus in the Wild" (reference repo wcy199705/DfFintheWild).

    from dffinthewild_amd import Network            # same call API as the reference's Network
    from dffinthewild_amd.Depth_Estimation_Network import Network   # same module name as the reference

Sub-modules: ``graph`` (layer table / weight contract), ``synth`` (deterministic synthetic weights
and stacks), ``engine`` (ctypes binding of libdffw.so), ``dist`` (batch sharding over the GPUs of
one node).  Importing ``engine`` (or ``Network``) requires the built HIP library; there is no
fallback path.
"""
__version__ = "0.1.0"


def __getattr__(name):
    if name == "Network":
        from .Depth_Estimation_Network import Network
        return Network
    raise AttributeError(name)
