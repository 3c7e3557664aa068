"""Event-handler base class (component C4, reference ``eventHandlers/EventHandler.py:1-26``).

A handler is a callable ``handler(eType, objName, eStatus)``; the watcher calls
every configured handler for every Job event.  Unknown event types go to
``errorEvent``; all four hooks default to a no-op.
"""
import logging

log = logging.getLogger(__name__)


class EventHandler:
    def __call__(self, eType, objName, eStatus):
        if eType == "ADDED":
            return self.addEvent(objName, eStatus)
        if eType == "MODIFIED":
            return self.modifEvent(objName, eStatus)
        if eType == "DELETED":
            return self.delEvent(objName, eStatus)
        return self.errorEvent(objName, eStatus)

    def addEvent(self, objName, eStatus):
        log.debug("%s: ADDED %s ignored", type(self).__name__, objName)

    def modifEvent(self, objName, eStatus):
        log.debug("%s: MODIFIED %s ignored", type(self).__name__, objName)

    def delEvent(self, objName, eStatus):
        log.debug("%s: DELETED %s ignored", type(self).__name__, objName)

    def errorEvent(self, objName, eStatus):
        log.debug("%s: unknown event for %s ignored", type(self).__name__, objName)
