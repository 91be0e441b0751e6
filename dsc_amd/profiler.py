"""`dsc.profile()` on top of dsc_traces_record / dsc_dump_traces / dsc_clear_traces (reference: python/dsc/profiler.py).

The dump is a Perfetto / chrome://tracing JSON file with two tracks: the operator calls on the host and the spans their
kernels took on the HIP stream.  The reference's stop_recording always starts a local web server and blocks until
ui.perfetto.dev has fetched the file; here the file is simply written (drag it into https://ui.perfetto.dev), and
`perfetto_url()` offers the served variant for interactive sessions."""
import contextlib
import functools
import http.server
import os
import threading

from . import _bindings as B
from .context import _get_ctx


def start_recording() -> None:
    B.dsc_traces_record(_get_ctx(), True)


def stop_recording(traces_file: str, clear: bool = True) -> str:
    ctx = _get_ctx()
    B.dsc_traces_record(ctx, False)
    B.dsc_dump_traces(ctx, os.fsencode(traces_file))
    if clear:
        B.dsc_clear_traces(ctx)
    return traces_file


@contextlib.contextmanager
def profile(dump_file: str = 'traces.json'):
    """with dsc.profile('run.json'): ...   — records every operator call inside the block."""
    start_recording()
    try:
        yield dump_file
    finally:
        stop_recording(dump_file, clear=True)


def perfetto_url(traces_file: str, port: int = 9001, one_shot: bool = True) -> str:
    """Serve `traces_file` on 127.0.0.1 for ui.perfetto.dev (CORS header set) from a daemon thread and return the URL to
    open.  With one_shot the server stops after the file has been fetched once."""
    directory, name = os.path.split(os.path.abspath(traces_file))

    class Handler(http.server.SimpleHTTPRequestHandler):
        def end_headers(self):
            self.send_header('Access-Control-Allow-Origin', 'https://ui.perfetto.dev')
            super().end_headers()

        def log_message(self, *args):
            pass

    server = http.server.ThreadingHTTPServer(('127.0.0.1', port), functools.partial(Handler, directory=directory))

    def run():
        if one_shot:
            server.handle_request()
            server.server_close()
        else:
            server.serve_forever()

    threading.Thread(target=run, daemon=True).start()
    return f'https://ui.perfetto.dev/#!/?url=http://127.0.0.1:{port}/{name}'
