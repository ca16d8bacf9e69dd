"""Ordered host-side prefetch for ``encode_batch_files``.

The reference feeds the encoder from a ``DataLoader`` with ``num_workers`` processes, one file per worker at a time
(audiotoken/datasets.py:107-139, core.py:244-267), so decoding / resampling overlaps the device work and all chunks of a
file stay in order. Here the same overlap comes from a small thread pool (WAV parsing, the resampling ``conv1d`` and
``numpy`` release the GIL): files are loaded ``num_workers`` at a time but YIELDED strictly in the given order, so the
segment stream — and therefore the batches, the token files and their append order — is identical to the sequential one.
Archives are streamed member by member by one background thread through a bounded queue (members are not random-access)."""
from __future__ import annotations

import queue
import threading
import weakref
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from typing import Callable, Iterable, Iterator, List, TypeVar

T = TypeVar("T")
R = TypeVar("R")
_END = object()


def ordered_map(fn: Callable[[T], R], items: Iterable[T], num_workers: int) -> Iterator[R]:
    """``map(fn, items)`` with up to ``num_workers`` calls in flight, results in input order. ``num_workers <= 0`` runs
    inline (exactly the sequential behaviour). Exceptions surface at the position of the item that raised."""
    if num_workers <= 0:
        for it in items:
            yield fn(it)
        return
    with ThreadPoolExecutor(max_workers=num_workers, thread_name_prefix="audiotoken-io") as pool:
        pending: deque = deque()
        it = iter(items)
        try:
            for item in it:
                pending.append(pool.submit(fn, item))
                if len(pending) >= num_workers:
                    yield pending.popleft().result()
            while pending:
                yield pending.popleft().result()
        finally:
            for f in pending:
                f.cancel()


def _put(q: "queue.Queue", stop: threading.Event, x) -> bool:
    while not stop.is_set():
        try:
            q.put(x, timeout=0.1)
            return True
        except queue.Full:
            continue
    return False


def _produce(q: "queue.Queue", stop: threading.Event, gen_fn) -> None:
    """Thread body of `background`. It holds the queue, the stop event and the generator factory — NOT the iterator object: a thread
    target bound to the object would keep it alive, so its __del__ could never run while the producer is blocked in put()."""
    gen = None
    try:
        gen = iter(gen_fn())
        for x in gen:
            if not _put(q, stop, x):
                return
        _put(q, stop, _END)
    except BaseException as e:  # delivered to the consumer
        _put(q, stop, e)
    finally:
        close = getattr(gen, "close", None)   # run the generator's finally blocks (tar / zip handles) in this thread
        if close is not None:
            try:
                close()
            except Exception:
                pass


class background(Iterator[T]):
    """Run a generator in one background thread, handing its items over through a bounded queue (order preserved).

    An iterator OBJECT, not a generator function: the thread starts when the object is made (so an archive streams ahead while
    earlier files are still being consumed), and ``close()`` / garbage collection / an exception in the consumer stops the producer —
    every ``put`` of the producer, including the final sentinel and a forwarded exception, gives up once ``stop`` is set, so the thread
    (and the tar / zip handle it holds) never outlives an abandoned consumer. The thread references only the queue, the event and the
    generator factory; ``weakref.finalize`` sets the event when the object is collected without ``close()``."""

    def __init__(self, gen_fn: Callable[[], Iterable[T]], depth: int = 8):
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        self._stop = threading.Event()
        self._done = False
        self._thread = threading.Thread(target=_produce, args=(self._q, self._stop, gen_fn), name="audiotoken-io-stream", daemon=True)
        self._finalizer = weakref.finalize(self, self._stop.set)
        self._thread.start()

    def __iter__(self):
        return self

    def __next__(self) -> T:
        if self._done:
            raise StopIteration
        x = self._q.get()
        if x is _END:
            self.close()
            raise StopIteration
        if isinstance(x, BaseException):
            self.close()
            raise x
        return x

    def close(self) -> None:
        self._done = True
        self._stop.set()

    def join(self, timeout: float = None) -> bool:
        """Wait for the producer thread to end (tests); True when it has."""
        self._thread.join(timeout)
        return not self._thread.is_alive()


def chunks_of_files(files: List[str], load_chunks: Callable[[str], list], num_workers: int) -> Iterator:
    """All chunks of all files, file order and chunk order preserved, ``num_workers`` files decoded ahead."""
    for chunk_list in ordered_map(load_chunks, files, num_workers):
        yield from chunk_list
