"""Small helpers shared by the facade modules (reference utils.py)."""
import time


def find_class_by_name(name, modules):
    """Searches the provided modules for the named class and returns it: the
    first module attribute called ``name``; StopIteration when there is none
    (reference utils.py:23-26 -- the plugin registry train.py:351-352 uses)."""
    found = [getattr(module, name, None) for module in modules]
    return next(a for a in found if a)


def exe_time(func):
    """Record function running time (reference utils.py:11-20)."""
    def timed(*args, **kwargs):
        t0 = time.time()
        back = func(*args, **kwargs)
        print("@%.3fs taken for {%s}" % (time.time() - t0, func.__name__))
        return back
    return timed


def get_local_time():
    return time.strftime("%y%m%d_%H%M%S", time.localtime())
