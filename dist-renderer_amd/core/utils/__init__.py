"""core.utils: like the reference's package (core/utils/__init__.py:3-6) it re-exports the public names of its modules."""
from core import _dropin

__path__ = _dropin.extend(__path__, __name__)

from .decoder_utils import *    # noqa: E402,F401,F403
from .loss_utils import *       # noqa: E402,F401,F403
from .train_utils import *      # noqa: E402,F401,F403
from .render_utils import *     # noqa: E402,F401,F403
