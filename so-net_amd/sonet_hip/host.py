"""Host-side hygiene for loops that launch a few hundred kernels per step.

``freeze_gc()`` -- call once the model, the optimizers and one warm-up step exist.  A process that has imported torch tracks ~270,000
container objects; CPython's cyclic collector walks ALL of them in a generation-2 collection: 70 ms on the MI355X host, once every 100-200
training steps (5.5 -> 5.2 ms per step at B = 64, and a 20-step forward window of 18 ms can be hit by one: tools/train_free_run.py,
profiles/r04y_gc_pause.log).  ``gc.freeze()`` moves what is alive now to the permanent generation: later collections walk only what was
created since (a step's garbage), and nothing is leaked -- the frozen objects are the model and the imported modules, which live as long as
the process anyway.  ``unfreeze_gc()`` undoes it."""
import gc


def freeze_gc():
    """Collect now, then exempt every live object from future cyclic collections.  Returns the number of frozen objects."""
    gc.collect()
    gc.freeze()
    return gc.get_freeze_count()


def unfreeze_gc():
    gc.unfreeze()
