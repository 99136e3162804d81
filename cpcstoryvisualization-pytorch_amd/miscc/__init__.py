from __future__ import division
from __future__ import print_function
